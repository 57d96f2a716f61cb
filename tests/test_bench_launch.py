"""bench.py's N-rank launch path: `python bench.py --gpus N` without a launcher around it must start N ranks
itself (as a child torch.distributed.run, before any GPU call) and print ONE JSON line with n_gpus = N."""
import importlib.util
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("rx_bench", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_starts_torchrun_child_with_same_arguments(monkeypatch):
    bench = _load_bench()
    seen = {}

    class R:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return R()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse()
    assert bench.self_launch(args) == 7  # the child's exit code is handed back
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    tail = cmd[cmd.index(os.path.join(ROOT, "bench.py")) + 1:]
    assert tail == ["--gpus", "4", "--steps", "3", "--warmup", "1"]
    assert seen["env"]["MASTER_ADDR"] == "127.0.0.1"


def test_main_self_launches_only_without_a_launcher(monkeypatch):
    """--gpus N with WORLD_SIZE unset -> self_launch and exit with its code; the parent makes no GPU call."""
    bench = _load_bench()
    calls = []
    monkeypatch.setattr(bench, "self_launch", lambda a: calls.append(a.gpus) or 0)
    monkeypatch.setattr(bench, "build_world", lambda a: (_ for _ in ()).throw(AssertionError("GPU path reached")))
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 0 and calls == [2]


@pytest.mark.gpu
@pytest.mark.parametrize("extra", [[], ["--no-graph"]], ids=["default", "eager"])
def test_bench_gpus_2_runs_two_ranks_on_one_gpu(extra):
    """The whole multi-rank control flow on the single GPU of a test box: RX_BENCH_BACKEND=gloo puts both ranks on
    device 0 and reduces through the host (never a reported number); rank 0 prints one line with n_gpus == 2."""
    env = dict(os.environ, RX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--settle", "1", "--bs", "16", "--ctx", "512", "--layers", "4", "--no-cpu-baseline", "--full-json"] + extra,
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["parallelism"] == "tp2"
    assert out["config"]["all_reduce"] == "gloo" and out["steps"] == 2
    assert len(out["roofline"]["per_rank_frac"]) == 2
    assert {"alone_us", "step_ms_overlap", "step_ms_no_overlap"} <= set(out["all_reduce"])
    # first contact: both ranks seen, the reduce checked against the fp32 sum before anything was timed
    fc = out["first_contact"]
    assert fc["ranks_seen"] == [0, 1] and fc["implementation"] == "gloo" and fc["all_reduce_check"]["ok_on_every_rank"]
    assert len(fc["can_access_peer"]) == 2
    # the other implementation (peer-to-peer two-shot kernel) ran as a fresh child job and reported its own timings
    leg = out["all_reduce"]["p2p_two_shot_leg"]
    assert "error" not in leg, leg
    assert leg["implementation"].startswith("p2p-two-shot") and leg["device_side_timeouts"] == 0
    assert leg["first_contact"]["ok_on_every_rank"] and {"alone_us", "step_ms_overlap", "step_ms_no_overlap"} <= set(leg)
    # the extend half of the metric is reported at N > 1 too: every rank's head shard, slowest rank's time
    assert "error" not in out["extend"], out["extend"]
    assert out["extend"]["tflops"] > 0 and out["extend"]["sharding"].startswith("tp2")


@pytest.mark.gpu
def test_bench_gpus_8_first_contact_on_one_gpu():
    """The target world size before the driver's SCALE run meets it: `python bench.py --gpus 8`
    self-launches eight ranks (all on device 0, gloo reducing through the host), every rank passes first_contact() --
    peer-access matrix, the all-reduce checked against the fp32 sum -- the TP = 8 shard (Hq 4 / Hkv 1 per rank) steps
    from HIP graphs or eagerly."""
    env = dict(os.environ, RX_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1",
                        "--settle", "1", "--bs", "16", "--ctx", "512", "--layers", "2", "--no-cpu-baseline", "--full-json",
                        "--no-custom-ar-leg"],   # (the two-shot kernel at world 8: tests/test_gpu_allreduce.py; a second 8-rank job here costs minutes)
                       capture_output=True, text=True, env=env, timeout=1500)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "tp8" and "Hq=4,Hkv=1" in out["config"]["workload"]
    assert len(out["roofline"]["per_rank_frac"]) == 8
    fc = out["first_contact"]
    assert fc["ranks_seen"] == list(range(8)) and fc["all_reduce_check"]["ok_on_every_rank"] and len(fc["can_access_peer"]) == 8
    assert out["extend"]["sharding"].startswith("tp8") and out["extend"]["tflops"] > 0
    # the compact line of an N > 1 run keeps the collective's figures
    c = _load_bench().compact_record(out)
    assert c["n_gpus"] == 8 and c["first_contact_ok"] and "alone_us" in c["all_reduce"] and len(json.dumps(c)) < 3000


@pytest.mark.gpu
def test_bench_tp_sim_shard_replays_from_hip_graphs():
    """One rank's shard of a TP=8 job (no collective) under graph replay: the step is launched as HIP-graph replays
    and the roofline kernel is still timed live."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--tp-sim", "8", "--steps", "3", "--warmup", "1",
                        "--settle", "1", "--bs", "64", "--ctx", "1024", "--layers", "8", "--no-cpu-baseline",
                        "--no-extend", "--no-radix-hit", "--no-extra"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    # default output: the full record on a '[bench-full] ' line, then ONE compact JSON line, last
    lines = r.stdout.splitlines()
    full = json.loads(next(ln for ln in lines if ln.startswith("[bench-full] "))[len("[bench-full] "):])
    assert full["n_gpus"] == 1 and full["config"]["step_launch"].startswith("hip-graph")
    assert full["roofline"]["launches"] == 3 and full["roofline"]["avg_launch_ms"] > 0
    assert lines[-1].startswith("{") and len(lines[-1]) < 2660
    out = json.loads(lines[-1])
    assert out["value"] == pytest.approx(full["value"], rel=1e-3) and out["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], abs=1e-4)
    assert out["config"]["parallelism"] == "tp8" and "tp_sim" in out["config"]


def test_compact_line_carries_both_halves_of_the_metric():
    """The final line of a default run: under 2.6 KB, the contract's keys, and -- as SCALAR keys of the
    `roofline` dict, right after `frac` / `traffic`, at most 24 keys in all (the driver's parser keeps scalars only and cut the
    tail in round 5) -- the extend half of the metric plus the legs the verdict sets targets on.  Fed with a committed full record."""
    bench = _load_bench()
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_bench_default_full.json")))
    c = bench.compact_record(full)
    line = json.dumps(c)
    assert len(line) < 2660, len(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in c, k
    assert set(c["config"]) >= {"workload", "global_batch", "seq_len", "parallelism"}
    rf = c["roofline"]
    assert len(rf) <= 24 and all(not isinstance(v, (dict, list)) for v in rf.values()), rf
    keys = list(rf)
    assert keys[:6] == ["bound", "achieved", "peak", "unit", "frac", "traffic"]
    assert keys[6:13] == ["extend_frac", "extend_kernel_tflops", "extend_tflops", "extend_ms_per_launch", "extend_kernel",
                          "extend_flops_per_launch", "extend_frac_at_sclk"]
    assert rf["bound"] == "hbm" and rf["frac"] == pytest.approx(full["roofline"]["frac"], abs=1e-4)
    assert rf["extend_frac"] == pytest.approx(full["extend"]["roofline"]["frac"], abs=1e-4)
    assert rf["extend_kernel_tflops"] == pytest.approx(full["extend"]["kernel_only"]["tflops"], rel=1e-3)
    assert rf["extend_frac"] == pytest.approx(rf["extend_flops_per_launch"] / (rf["extend_ms_per_launch"] * 1e-3) / 1e12 / 2500.0, rel=1e-3)
    assert rf["mla_decode_fp8_op_frac"] == pytest.approx(full["mla_decode"]["fp8_rows"]["op_frac_of_hbm_peak"], abs=1e-4)
    assert rf["tp8_kernel_frac"] == pytest.approx(full["extra"]["tp_sim"]["tp8"]["kernel_frac_of_hbm_peak"], abs=1e-4)
    for k in ("config3_70b_tp8_shard_kernel_frac", "config1_decode_kernel_frac", "deterministic_unified_frac", "rel_bias_1024_frac",
              "prefill2k_kernel_frac"):
        assert k in rf, k
    assert c["more"]["d64_frac"] == pytest.approx(full["extend"]["other_head_dims"]["d64"]["frac"], abs=1e-4)
    assert c["cpu_baseline"]["kind"] in ("reference", "port") and c["cpu_baseline"]["cores"] >= 1
    # an errored leg must not take the line down
    full["extend"] = {"error": "boom"}
    assert bench.compact_record(full)["roofline"]["extend_error"] == "boom"
    # nor a record that misses keys the compact form indexes: emit() falls back to a minimal line
    import contextlib
    import io

    broken = {k: v for k, v in full.items() if k != "config"}
    buf = io.StringIO()

    class A:
        full_json = False

    with contextlib.redirect_stdout(buf):
        bench.emit(broken, A)
    last = json.loads(buf.getvalue().splitlines()[-1])
    assert last["value"] == full["value"] and "compact_record_error" in last and last["roofline"]["frac"] == full["roofline"]["frac"]
