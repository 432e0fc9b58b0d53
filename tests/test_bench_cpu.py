"""bench.py launches its own ranks when it is not under torchrun: `python bench.py --gpus 2` must spawn two processes
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*), run the sharded entry point with one all-gather per global batch and print
ONE JSON line with n_gpus 2.  Exercised on the CPU with the script's stub engine over gloo (--stub)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env_extra=None):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(env_extra or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--stub", "--steps", "2", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip() and not ln.startswith("[Gloo]")]    # (gloo's own connect banner)
    assert len(lines) == 1, r.stdout                      # stdout carries the ONE JSON line (rank 0 only)
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks():
    d = _run(["--gpus", "2"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["value"] > 0 and d["scaling"] == "weak"
    assert d["config"]["global_batch"] == 8               # 4 images per rank


def test_bench_sharded_global_batch_smaller_than_world_times_micro():
    d = _run(["--gpus", "2", "--global-batch", "3"])      # ragged shards (2 + 1): still one line, strong scaling
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["global_batch"] == 3


def test_bench_single_rank_stub():
    d = _run([])
    assert d["n_gpus"] == 1 and d["value"] > 0


def test_bench_cites_the_final_collection_of_the_newest_round(tmp_path, monkeypatch):
    """`roofline.traffic` comes from profiles/rNN_traffic.json of the newest round — the round's FINAL collection, not an earlier one
    of the same round (rNNa_ / rNNe_ sort after rNN_ alphabetically and were picked up once)"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    prof = tmp_path / "profiles"
    prof.mkdir()
    for name, val in (("r05_traffic.json", 1.0), ("r06_traffic.json", 2.0), ("r06a_traffic.json", 3.0), ("r06e_traffic.json", 4.0)):
        (prof / name).write_text(json.dumps({"head": name, "conv3": {"hbm_bytes_per_launch": val}}))
    (prof / "r06_traffic_inflight_plans.json").write_text(json.dumps({"conv3": {"hbm_bytes_per_launch": 5.0}}))
    (prof / "r06e_traffic_inflight_plans.json").write_text(json.dumps({"conv3": {"hbm_bytes_per_launch": 6.0}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    assert bench.measured_traffic() == (2.0, "r06_traffic.json @ r06_traffic.json")
    assert bench.measured_traffic_inflight_plans()[0] == 5.0
