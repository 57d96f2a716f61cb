"""plugin.register() executed against the reference's own attention registry (attention_registry.py:28-39,
server_args.py:386-387) and the backend built through the registered factory from a reference-shaped runner.
Runs where /root/reference exists (the build container); the GPU box has no reference tree."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/python/sglang"), reason="needs the reference tree")
def test_register_against_reference_registry_and_build_backend():
    env = dict(os.environ, HIP_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "check_plugin_register.py")],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    res = json.loads(line[len("RESULT "):])
    assert res["choice_added"] and res["factory_added"] and res["others_kept"]
    assert res["isinstance_abc"] and res["cls"] == "SGLangHipRadixAttnBackend" and res["abstract_left"] == []
    assert res["hooks_are_ours"]
    # TP=2 of 32 / 8 heads, page 16, splits from server_args (triton_backend.py:140-170)
    assert (res["num_head"], res["num_kv_head"], res["page_size"], res["max_kv_splits"]) == (16, 4, 16, 8)
