"""The commands the driver runs, as subprocesses on the GPU box: `python bench.py` (N = 1), `python bench.py --gpus N`
(bare: one process, a sharded instance; here with two shards on device 0 through PHYHIP_BENCH_DEVICES) and the
torch.distributed.run form (one process per GPU; here one rank).  Each must end its stdout with ONE JSON line that carries
the contract's fields."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(cmd, env_extra, timeout=600):
    env = dict(os.environ)
    env.update(env_extra)
    r = subprocess.run(cmd, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=timeout)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [x for x in r.stdout.splitlines() if x.strip()]
    assert lines, r.stderr[-2000:]
    return json.loads(lines[-1]), lines  # the JSON line is the LAST line of stdout


CONTRACT = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config")


def _detail():
    """everything the run measured (bench.py writes it next to itself; the printed line is its compact form)"""
    return json.load(open(os.path.join(ROOT, "bench_detail.json")))


def _physical(r):
    """a roofline block of the printed line: physical figures only -- nothing above 1, and reproducible from its own keys"""
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert 0.0 < r["frac"] < 1.0 and abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-4
    assert abs(r["achieved"] - r["traffic"] / (r["kernel_avg_us"] * 1e-6) / 1e9) < 2e-3 * r["achieved"]
    assert r["traffic_is"] in ("pmc", "model") and r["frac_algorithmic"] > r["frac"]


def test_default_single_gpu_line():
    d, lines = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"], {})
    assert len(lines) == 1 and len(lines[0]) < 2000  # N = 1: nothing but the JSON line on stdout, short enough for a log tail
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "none" and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["config"]["workload"] == "cfg2_nt_100x50k" and d["lnL_rel_err"] < 1e-6
    r = d["roofline"]
    _physical(r)
    # the kernel rocprofv3 shows for this launch: the two-wave-shape list form with in-step tip x tip children
    assert r["kernel"] == "traverse_nt2_mixed_kernel<4, true>"
    # (the all-stored companion: a slower KERNEL; three timed steps are too few to order the two whole-step throughputs reliably)
    assert 0 < r["virtual_buffers"] < 98 and r["all_stored_kernel_us"] > r["kernel_avg_us"] and r["all_stored_value"] > 0
    assert r["materialise_after_traversal_us"] > 0.0
    x = d["cfg3_aa_200x10k"]
    assert x["lnL_rel_err"] < 1e-6 and x["kernel"].startswith("traverse_aa_kernel<4") and 0.0 < x["frac"] < 1.0 and 0.0 < x["mfma_frac"] < 1.0
    y = d["cfg4_nt_100x1M_one_gpu"]
    assert y["lnL_rel_err"] < 1e-6 and 0.0 < y["frac"] < 1.0 and y["value"] > 0
    c = d["call_us"]
    for k in ("spr_500x100k", "dlk_500x100k", "chain_1eig_5dlk_500x100k", "spr_54x382", "call_54x382", "spr_37x429_aa", "call_37x429_aa"):
        assert c[k] > 0.0, k
    full = _detail()
    assert full["input_checksum_ok"] and full["value"] == pytest.approx(d["value"], rel=1e-4)
    fr = full["roofline"]
    assert fr["all_buffers_stored"]["lnL"] == full["lnL"]   # the same double, stored or not
    m = fr["materialise"]
    assert m["virtual_before_after_materialised"][0] > 0 and m["virtual_before_after_materialised"][1] == 0
    assert m["first_short_call_after_traversal_us"] > m["same_call_repeated_us"]
    assert full["extra"]["call_latency"]["spr_37x429_aa"]["finite"]


def test_bare_multi_gpu_command_with_two_shards_on_device_0():
    d, _ = _run([sys.executable, "bench.py", "--gpus", "2", "--patterns", "200000", "--steps", "3", "--warmup", "1"],
                {"PHYHIP_BENCH_DEVICES": "0,0"})
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert d["config"]["rccl_ranks"] == 1            # two shards on ONE device share its (one-rank) communicator
    assert d["config"]["patterns_per_gpu"] == 100000
    pr = d["per_rank"]
    assert pr["comm_size"] == [1] and pr["kernel_us"][0] > 0 and pr["collective_us"][0] > 0
    full = _detail()
    ss = full["strong_scaling"]
    assert abs(full["lnL"] - ss["single_gpu_lnL"]) / abs(full["lnL"]) < 1e-10   # the sharded sum against the whole alignment on one GPU
    assert d["value"] > 0 and d["strong_scaling"]["single_gpu_value"] > 0


def test_torchrun_form_one_rank():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1", "--patterns", "200000", "--steps", "3", "--warmup", "1"]
    d, _ = _run(cmd, {"PHYHIP_BENCH_FORCE_DIST": "1"})
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["config"]["rccl_ranks"] == 1
    pr = d["per_rank"]
    assert pr["comm_size"] == [1] and pr["patterns"] == [200000] and pr["kernel_us"][0] > 0 and pr["collective_us"][0] > 0
    full = _detail()
    assert "phyhip_comm_init_rank" in full["config"]["mode"]
    ss = full["strong_scaling"]
    assert abs(full["lnL"] - ss["single_gpu_lnL"]) / abs(full["lnL"]) < 1e-10


def test_full_size_sharded_line_matches_the_reference_shard_sum():
    """cfg4 at its full 1 M patterns, eight shards on device 0: lnL against the reference's own shard sum (gate 1e-6)."""
    d, _ = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extra"],
                {"PHYHIP_BENCH_DEVICES": "0,0,0,0,0,0,0,0"}, timeout=900)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["patterns_total"] == 1000000
    assert d["lnL_rel_err"] < 1e-6
