#!/usr/bin/env python3
"""Prints the two digests selfrec_amd/dropin.py keeps per fusable reference model file: SHA-256 of the bytes and of the
canonical syntax dump (comments, blank lines, indentation style and docstrings do not enter it).  Run in the build container:
    python tests/golden/make_fusable_digests.py /root/reference
Facts about the reference's files, not copies of them."""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from selfrec_amd.dropin import syntax_digest  # noqa: E402

root = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"
for name in ("XSimGCL", "LightGCN", "SimGCL", "SGL", "MF"):
    src = open(os.path.join(root, "model", "graph", name + ".py"), "rb").read()
    print(f'    "{name}": ("{hashlib.sha256(src).hexdigest()}",\n               "{syntax_digest(src.decode())}"),')
