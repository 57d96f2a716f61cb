#!/usr/bin/env python3
"""Print an RX_DUMP_DIR argument dump (csrc/rx_common.h: dump_on_error): the .txt record as it stands and the .bin
parameter struct field by field through the ctypes binding.     python tools/decode_dump.py <dir>/rx_extend_attn_<pid>_<n>"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sglang_amd import lib  # noqa: E402


def fields(obj, indent=""):
    for name, tp in obj._fields_:
        v = getattr(obj, name)
        if hasattr(v, "_fields_"):
            print(f"{indent}{name}:")
            fields(v, indent + "  ")
        else:
            print(f"{indent}{name} = {hex(v) if isinstance(v, int) and tp in (C.c_void_p,) and v else v}")


def main(stem):
    stem = stem[:-4] if stem.endswith((".txt", ".bin")) else stem
    print(open(stem + ".txt").read())
    if not os.path.exists(stem + ".bin"):
        return
    raw = open(stem + ".bin", "rb").read()
    cls = lib.RxExtendParams if "extend_attn" in os.path.basename(stem) else lib.RxDecodeParams
    if len(raw) != C.sizeof(cls):
        print(f"(parameter struct of {len(raw)} bytes, binding has {C.sizeof(cls)}: another ABI version)")
        return
    fields(cls.from_buffer_copy(raw))


if __name__ == "__main__":
    main(sys.argv[1])
