"""File logger with the interface of reference util/logger.py:5-17 (``Log(module, filename)``,
``.add(text)``), writing to ``./log/<filename>.log``."""
import logging
import os


class Log:
    directory = './log/'

    def __init__(self, module, filename):
        os.makedirs(self.directory, exist_ok=True)
        self.logger = logging.getLogger(module)
        self.logger.setLevel(logging.INFO)
        sink = logging.FileHandler(os.path.join(self.directory, filename + '.log'))
        sink.setFormatter(logging.Formatter('%(asctime)s - %(name)s - %(levelname)s - %(message)s'))
        self.logger.addHandler(sink)

    def add(self, text):
        self.logger.info(text)
