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


def test_default_single_gpu_line():
    d, lines = _run([sys.executable, "bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--no-call-latency"], {})
    assert len(lines) == 1  # N = 1: nothing but the JSON line on stdout
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "none" and d["dtype"] == "f64" and d["higher_is_better"] is True
    assert d["config"]["workload"].startswith("cfg2_nt_100x50k")
    assert d["lnL_rel_err"] < 1e-6 and d["input_checksum_ok"]
    r = d["roofline"]
    # `frac` is SURVEY 8(d)'s ALGORITHMIC figure (it also charges what the launch keeps in registers or leaves virtual: it passes 1
    # and says so); the physical one -- counter bytes of these very kernel sources, or null -- can not
    assert r["bound"] == "hbm" and 0.0 < r["frac"] < 2.0 and r["kernel"] == "traverse_nt2_kernel" and "algorithmic" in r["frac_is"]
    assert r["frac_real"] is None or 0.0 < r["frac_real"] < 1.0
    vb = r["virtual_buffers"]
    assert 0 < vb["virtual_after_launch"] < vb["internal_buffers"] and r["all_buffers_stored"]["kernel_avg_us"] > r["kernel_avg_us"]
    assert r["all_buffers_stored"]["lnL"] == d["lnL"]   # the same double, stored or not
    x = d["extra"]["cfg3_aa_200x10k"]
    assert x["lnL_rel_err"] < 1e-6 and x["roofline"]["kernel"] == "traverse_aa_kernel" and 0.0 < x["roofline"]["frac"] < 2.0
    assert x["roofline"]["frac_real"] is None or 0.0 < x["roofline"]["frac_real"] < 1.0


def test_bare_multi_gpu_command_with_two_shards_on_device_0():
    d, _ = _run([sys.executable, "bench.py", "--gpus", "2", "--patterns", "200000", "--steps", "3", "--warmup", "1"],
                {"PHYHIP_BENCH_DEVICES": "0,0"})
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["steps"] == 3
    assert d["config"]["rccl_ranks"] == 1            # two shards on ONE device share its (one-rank) communicator
    assert d["config"]["patterns_per_gpu"] == 100000
    ss = d["strong_scaling"]
    assert abs(d["lnL"] - ss["single_gpu_lnL"]) / abs(d["lnL"]) < 1e-10   # the sharded sum against the whole alignment on one GPU
    assert d["value"] > 0 and ss["single_gpu_value"] > 0


def test_torchrun_form_one_rank():
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "1", "--patterns", "200000", "--steps", "3", "--warmup", "1"]
    d, _ = _run(cmd, {"PHYHIP_BENCH_FORCE_DIST": "1"})
    for k in CONTRACT:
        assert k in d, k
    assert d["n_gpus"] == 1 and d["scaling"] == "strong"
    assert d["config"]["rccl_ranks"] == 1 and "phyhip_comm_init_rank" in d["config"]["mode"]
    ss = d["strong_scaling"]
    assert abs(d["lnL"] - ss["single_gpu_lnL"]) / abs(d["lnL"]) < 1e-10


def test_full_size_sharded_line_matches_the_reference_shard_sum():
    """cfg4 at its full 1 M patterns, eight shards on device 0: lnL against the reference's own shard sum (gate 1e-6)."""
    d, _ = _run([sys.executable, "bench.py", "--gpus", "8", "--steps", "2", "--warmup", "1", "--no-extra"],
                {"PHYHIP_BENCH_DEVICES": "0,0,0,0,0,0,0,0"}, timeout=900)
    assert d["n_gpus"] == 8 and d["scaling"] == "strong" and d["config"]["patterns_total"] == 1000000
    assert d["lnL_rel_err"] < 1e-6
