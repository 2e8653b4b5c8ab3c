#!/usr/bin/env python3
"""The source stamp of libnfx.so: the first 16 hex digits of a sha-256 over csrc's *.hip, *.cpp and *.h (name, then bytes, in name order)
and include/nfx.h.  ONE definition, used three times: the Makefile compiles it into the library (nfx_build_stamp()), bench.py compares
that string with this function's value over the sources on disk and refuses to run on a mismatch, tests/test_abi.py asserts the same."""
import hashlib
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def source_stamp(csrc_dir: str = HERE) -> str:
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc_dir)):
        if fn.endswith((".hip", ".h", ".cpp")):
            h.update(fn.encode())
            with open(os.path.join(csrc_dir, fn), "rb") as fh:
                h.update(fh.read())
    hdr = os.path.join(csrc_dir, "..", "..", "include", "nfx.h")
    h.update(b"include/nfx.h")
    with open(hdr, "rb") as fh:
        h.update(fh.read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(source_stamp())
