STAMPS=1 VARIANTS=0 python tools/ext32_ab.py 2>&1 | tail -5
ZERO=1 STAMPS=1 VARIANTS=0 python tools/ext32_ab.py 2>&1 | tail -5
