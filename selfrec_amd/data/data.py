"""Root data holder (reference data/data.py:1-5): config + raw train/test lists."""


class Data:
    def __init__(self, conf, training, test):
        self.config = conf
        self.training_data = training
        self.test_data = test
