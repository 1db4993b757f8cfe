"""Dataset files -> in-memory triples, in the formats of reference data/loader.py:22-41.

Graph data: one interaction per line, ``user item weight`` separated by single spaces,
returned as ``[[user_str, item_str, float_weight], ...]`` in file order (the order matters:
ids are assigned by first appearance, reference data/ui_graph.py:29-38).
"""
import os


class FileIO:
    @staticmethod
    def load_data_set(file, rec_type):
        if rec_type == 'graph':
            triples = []
            with open(file) as src:
                for raw in src:
                    parts = raw.strip().split(' ')
                    triples.append([parts[0], parts[1], float(parts[2])])
            return triples
        if rec_type == 'sequential':
            sequences = {}
            with open(file) as src:
                for raw in src:
                    key, _, tail = raw.strip().partition(':')
                    sequences[key] = tail.split()
            return sequences
        raise ValueError(f"unknown recommender type {rec_type!r}")

    @staticmethod
    def write_file(dir, file, content, op='w'):
        os.makedirs(dir, exist_ok=True)
        with open(dir + file, op) as dst:
            dst.writelines(content)

    @staticmethod
    def delete_file(file_path):
        if os.path.exists(file_path):
            os.remove(file_path)
