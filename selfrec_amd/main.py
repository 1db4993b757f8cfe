"""Non-interactive launcher (the reference's main.py:10-34 asks for the model on stdin):

    python -m selfrec_amd.main XSimGCL [--conf conf/XSimGCL.yaml] [--synthetic yelp2018]

``--synthetic SHAPE`` writes a generated dataset of that shape (selfrec_amd/synth.py) to the
paths the config names, if they do not exist yet -- the reference's dataset files are not
redistributable.
"""
import argparse
import os
import time

from . import synth
from .SELFRec import SELFRec
from .util.conf import ModelConf

MODELS = ['MF', 'LightGCN', 'XSimGCL', 'SimGCL', 'SGL', 'DirectAU', 'MixGCF', 'BUIR', 'SelfCF']


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument('model', choices=MODELS)
    ap.add_argument('--conf', default=None)
    ap.add_argument('--synthetic', default=None, choices=sorted(synth.SHAPES))
    args = ap.parse_args(argv)
    conf = ModelConf(args.conf or f'./conf/{args.model}.yaml')
    if args.synthetic and not os.path.exists(conf['training.set']):
        tu, ti, su, si, _, _ = synth.make_dataset(args.synthetic)
        os.makedirs(os.path.dirname(conf['training.set']) or '.', exist_ok=True)
        synth.write_text(conf['training.set'], tu, ti)
        synth.write_text(conf['test.set'], su, si)
    t0 = time.time()
    SELFRec(conf).execute()
    print(f"Running time: {time.time() - t0:.2f} s")


if __name__ == '__main__':
    main()
