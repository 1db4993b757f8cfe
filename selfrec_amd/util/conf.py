"""Configuration object with the access protocol of reference util/conf.py:5-28.

``conf['key']`` returns the value of a flat (dotted) key or of a nested per-model block
and terminates the process when the key is absent (the reference prints a message and calls
``exit(-1)``; the same message and exit status are kept so wrapper scripts behave
identically); ``conf.contain(key)`` tests for presence.  Additionally accepts a ready-made
dict so benchmarks and tests do not need a YAML file on disk.
"""
import os

import yaml


class ModelConf:
    def __init__(self, source):
        if isinstance(source, dict):
            self.config = dict(source)
        else:
            self.config = self._parse(source)

    @staticmethod
    def _parse(path):
        if not os.path.exists(path):
            print('Config file is not found!')
            raise IOError(path)
        with open(path, 'r') as handle:
            try:
                loaded = yaml.safe_load(handle)
            except yaml.YAMLError as err:
                print(f"Error in configuration file: {err}")
                raise IOError(path) from err
        return loaded or {}

    def contain(self, key):
        return key in self.config

    def __getitem__(self, key):
        try:
            return self.config[key]
        except KeyError:
            print('Parameter ' + key + ' is not found in the configuration file!')
            raise SystemExit(-1)

    def get(self, key, default=None):
        """Extension: optional keys (e.g. ``seed``, ``world.size``) default to the
        reference behaviour when absent."""
        return self.config.get(key, default)

    # kept for code that calls the reference's method name directly
    def read_configuration(self, path):
        self.config = self._parse(path)
