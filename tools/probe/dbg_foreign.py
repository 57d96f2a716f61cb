import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, torch
import parity_util as parity
orig = parity.check_out
def dbg(got, want, dtype, tag=None, ulps=1.0, absw=None):
    got = np.asarray(got, dtype=np.float64); want = np.asarray(want, dtype=np.float64)
    d = np.abs(got - want)
    bad = d > 0.05
    print("CHECK", tag, str(dtype), "max", d.max(), "nbad", int(bad.sum()), "nan", int(np.isnan(got).sum()))
    if bad.any():
        rows = np.unique(np.nonzero(bad)[0]); heads = np.unique(np.nonzero(bad)[1])
        print("  bad rows", rows.tolist()[:50], "heads", heads.tolist())
        r = rows[0]; print("  got", got[r, heads[0], :8], "want", want[r, heads[0], :8])
    return 0.0
parity.check_out = dbg
import test_foreign_pool as t
for dt in (torch.float16, torch.bfloat16):
    for ps in (1, 16):
        for mode in ("0", "2"):
            os.environ["RX_EXT_PW"] = mode
            print("=== dtype", dt, "ps", ps, "pw", mode)
            try:
                t.test_extend_and_decode_through_a_foreign_reference_shaped_pool.__wrapped__(ps, dt) if hasattr(t.test_extend_and_decode_through_a_foreign_reference_shaped_pool, "__wrapped__") else t.test_extend_and_decode_through_a_foreign_reference_shaped_pool(ps, dt)
            except Exception as e:
                print("EXC", type(e).__name__, str(e)[:300])
