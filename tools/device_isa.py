#!/usr/bin/env python3
"""Disassembly of the gfx950 kernels INSIDE the built library (selfrec_amd/lib/libselfrec_hip.so): the code objects are
cut out of the .hip_fatbin section (one clang offload bundle per translation unit) and run through llvm-objdump.

    python tools/device_isa.py [kernel-name-substring]        prints the matching kernels' instructions

`kernels(path)` -> {demangled kernel name: [instruction lines]} is what tests/test_isa_async_lds.py checks: the
hand-placed asynchronous LDS reads of nce_tile_f32 must never have their destination registers read before the
s_waitcnt that settles them (a compiler-made copy there copies stale bytes -- found the hard way, round 5)."""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "selfrec_amd", "lib", "libselfrec_hip.so")


def code_objects(path=LIB, arch="gfx950"):
    """the device ELF images for `arch` held by the shared library, one per translation unit"""
    blob = open(path, "rb").read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if arch in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + 1)
    return out


def kernels(path=LIB, arch="gfx950", name_filter=""):
    """{demangled name: [instruction text, ...]} of every function symbol in the library's device code"""
    found = {}
    for image in code_objects(path, arch):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(image)
            f.flush()
            text = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", "-C", f.name],
                                  capture_output=True, text=True, check=True).stdout
        cur = None
        for line in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.*)>:$", line)
            if m:
                cur = m.group(1)
                if name_filter in cur:
                    found.setdefault(cur, [])
                else:
                    cur = None
                continue
            if cur is not None and line.startswith("\t"):
                ins = line.split("//")[0].strip()
                if ins:
                    found[cur].append(ins)
    return found


def _regs(operand):
    """VGPR numbers named by one operand ('v12', 'v[4:7]'); empty for anything else"""
    m = re.fullmatch(r"v(\d+)", operand)
    if m:
        return {int(m.group(1))}
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    return set()


def async_lds_violations(instructions):
    """Instructions that READ (or overwrite) a VGPR while a ds_read into it may still be in flight.

    Model: LDS and scalar-memory operations enter one in-order queue (the lgkm counter); `s_waitcnt lgkmcnt(N)` settles
    all but the N youngest.  Any instruction naming a VGPR that a queued `ds_read*` still owes -- as a source or as a
    destination -- is a violation.  (Scalar loads may return out of order, which only matters for N > 0 with scalar loads
    in the queue; the compiler waits for those with N = 0.)  Branch targets are not followed: the kernels checked keep
    each read and its wait in one straight-line region, and a label in between settles nothing -- the strict side."""
    queue, bad = [], []                      # queue: [set of VGPRs owed] per outstanding lgkm operation, oldest first
    for k, ins in enumerate(instructions):
        op, _, rest = ins.partition(" ")
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", rest)
            if m:
                keep = int(m.group(1))
                queue = queue[max(0, len(queue) - keep):] if keep else []
            continue
        ops = [o.strip() for o in rest.split(",")] if rest else []
        named = set()
        for o in ops:
            named |= _regs(o.split(" ")[0])
        owed = set().union(*queue) if queue else set()
        if op.startswith("ds_read"):
            dst = _regs(ops[0]) if ops else set()
            if named & owed:
                bad.append((k, ins, sorted(named & owed)))
            queue.append(dst)
            continue
        if named & owed:
            bad.append((k, ins, sorted(named & owed)))
        if op.startswith("ds_") or op.startswith("s_load") or op.startswith("s_buffer_load"):
            queue.append(set())
    return bad


if __name__ == "__main__":
    flt = sys.argv[1] if len(sys.argv) > 1 else ""
    for name, body in kernels(name_filter=flt).items():
        print(f"== {name}: {len(body)} instructions")
        if "-v" in sys.argv:
            print("\n".join(body))
        for k, ins, regs in async_lds_violations(body):
            print(f"   pending-register use at {k}: {ins}    (v{regs})")
