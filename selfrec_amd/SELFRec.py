"""Loader + dispatcher with the contract of reference SELFRec.py:4-25:
``SELFRec(conf).execute()`` reads the train/test files named by the config and runs
``model.<type>.<name>.<name>(conf, training_set, test_set).execute()``."""
import importlib

from .data.loader import FileIO


class SELFRec:
    def __init__(self, config):
        self.config = config
        kind = config['model']['type']
        # lazy, natively parsed triples: behave like the reference's lists for any model that touches them
        self.training_data = FileIO.open_data_set(config['training.set'], kind)
        self.test_data = FileIO.open_data_set(config['test.set'], kind)
        self.kwargs = {}
        print('Reading data and preprocessing...')

    def execute(self):
        kind, name = self.config['model']['type'], self.config['model']['name']
        module = importlib.import_module(f"{__package__}.model.{kind}.{name}")
        getattr(module, name)(self.config, self.training_data, self.test_data, **self.kwargs).execute()
