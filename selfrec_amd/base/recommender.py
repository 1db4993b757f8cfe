"""Root template of every recommender.

Interface contract (what model classes written against SELFRec rely on, reference
base/recommender.py:7-83): the constructor signature ``(conf, training_set, test_set, **kwargs)``,
the attribute names set from the config (``emb_size``, ``maxEpoch``, ``batch_size``, ``lRate``,
``reg``, ``ranking``, ``output``, ``model_log``, ``result``, ``recOutput``), the seven overridable
hooks and the ``execute()`` stage order build -> train -> test -> evaluate with the same console
lines.  The implementation is table-driven: the config keys, the banner and the run stages are data.
"""
import os
import time

from ..data.data import Data
from ..util.logger import Log

# attribute name -> (config key, converter)
_CONF_FIELDS = (
    ("ranking", "item.ranking.topN", None),
    ("emb_size", "embedding.size", int),
    ("maxEpoch", "max.epoch", int),
    ("batch_size", "batch.size", int),
    ("lRate", "learning.rate", float),
    ("reg", "reg.lambda", float),
    ("output", "output", None),
)

# console banner: label -> how to obtain the value from the instance
_BANNER = (
    ("Model:", lambda s: s.model_name),
    ("Training Set:", lambda s: os.path.abspath(s.config["training.set"])),
    ("Test Set:", lambda s: os.path.abspath(s.config["test.set"])),
    ("Embedding Dimension:", lambda s: s.emb_size),
    ("Maximum Epoch:", lambda s: s.maxEpoch),
    ("Learning Rate:", lambda s: s.lRate),
    ("Batch Size:", lambda s: s.batch_size),
    ("Regularization Parameter:", lambda s: s.reg),
)


def _noop(self, *args, **kwargs):
    return None


class Recommender:
    # hooks a model overrides (all default to doing nothing, like the reference's)
    build = train = test = save = load = _noop

    def predict(self, u):
        return None

    def evaluate(self, rec_list):
        return None

    def __init__(self, conf, training_set, test_set, **kwargs):
        self.config = conf
        self.data = Data(conf, training_set, test_set)
        self.model_name = conf["model"]["name"]
        for attr, key, convert in _CONF_FIELDS:
            raw = conf[key]
            setattr(self, attr, convert(raw) if convert else raw)
        started = time.strftime("%Y-%m-%d %H-%M-%S", time.localtime(time.time()))
        self.model_log = Log(self.model_name, "%s %s" % (self.model_name, started))
        self.result, self.recOutput = [], []

    def initializing_log(self):
        log = self.model_log
        log.add("### model configuration ###")
        for key, value in self.config.config.items():
            log.add("%s=%s" % (key, value))

    def print_model_info(self):
        for label, value_of in _BANNER:
            print(label, value_of(self))
        if self.config.contain(self.model_name):
            section = self.config[self.model_name]
            print("Specific parameters:", "  ".join("%s:%s" % (k, section[k]) for k in section))

    def execute(self):
        self.initializing_log()
        self.print_model_info()
        rec_list = None
        for banner, stage in (("Initializing and building model...", "build"), ("Training Model...", "train"),
                              ("Testing...", "test"), ("Evaluating...", "evaluate")):
            print(banner)
            if stage == "evaluate":
                self.evaluate(rec_list)
            elif stage == "test":
                rec_list = self.test()
            else:
                getattr(self, stage)()
