"""Root data holder (interface of reference data/data.py:1-5): keeps the config and the raw train / test
triples under the attribute names every subclass and model reads."""


class Data:
    _FIELDS = ("config", "training_data", "test_data")

    def __init__(self, conf, training, test):
        # test may equally be a validation split: nothing here depends on which
        for name, value in zip(self._FIELDS, (conf, training, test)):
            setattr(self, name, value)
