"""Virtual buffers, the host side: a property test of the queue rewriting of the BUILT library (phyhip_queue.hip: rewrite_pending,
devirtualise*) on the CPU.  tests/virt_model.hip links libphyhip.so, feeds a hand-built Instance random queues, evaluations, readers
and matrix / tip changes, and follows every rewritten queue with the kernels' forwarding rules on symbolic values, in lockstep with
the plain semantics (every operation stored, in queue order).  No device is touched: the device side of the same contract is
tests/test_gpu_virtual.py.

The second test shows that the model sees what it is there for: the library's bookkeeping rebuilt with the defect a real PhyML
search found in round 5 (a short launch that rewrites a virtual buffer left its old definition in force) fails it within a few
hundred launches."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "bin", "hipcc")
LIBDIR = os.path.join(ROOT, "phyml_amd", "lib")
CSRC = os.path.join(ROOT, "phyml_amd", "csrc")

pytestmark = pytest.mark.skipif(not (os.path.exists(HIPCC) and os.path.exists(os.path.join(LIBDIR, "libphyhip.so"))),
                                reason="needs hipcc and the built library")

HOST_ONLY = ["--offload-arch=gfx950", "-std=c++17", "-O1", "--cuda-host-only", "-w"]


def compile_host(src, obj, cwd):
    subprocess.run([HIPCC] + HOST_ONLY + ["-c", "-o", obj, src], check=True, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


def link(objs, exe, cwd):
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-w", "-o", exe] + objs + ["-L" + LIBDIR, "-lphyhip", "-Wl,-rpath," + LIBDIR],
                   check=True, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.PIPE)


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    d = str(tmp_path_factory.mktemp("virt_model"))
    obj = os.path.join(d, "virt_model.o")
    compile_host(os.path.join(ROOT, "tests", "virt_model.hip"), obj, d)
    exe = os.path.join(d, "virt_model")
    link([obj], exe, d)
    return d, obj, exe


def run(exe, seed, events, tips, soa, in_step=1):
    return subprocess.run([exe, str(seed), str(events), str(tips), str(soa), str(in_step)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                          timeout=300)


@pytest.mark.parametrize("soa", [1, 0], ids=["nt_two_lane", "aa"])
def test_rewritten_queues_keep_the_plain_semantics(model, soa):
    _, _, exe = model
    base = int(os.environ.get("PHYHIP_FUZZ_SEED", "0"))
    for seed, tips in ((1, 12), (2, 5), (3, 30), (4, 8), (5, 16), (6, 3)):
        r = run(exe, base + seed, 3000, tips, soa)
        assert r.returncode == 0 and "VIRT_MODEL OK" in r.stdout, (seed, tips, r.stdout[-300:], r.stderr[-600:])
        # ... and the run did exercise the machinery
        f = r.stdout.split(":")[1].split(",")
        launches, skipped, in_step = (int(x.split()[0]) for x in f[:3])
        assert launches > 1000 and skipped > 50 and in_step > 100, r.stdout


def test_without_in_step_children(model):
    """Every virtual child re-issued as a non-storing step of its own (the diag build's PHYHIP_VIRT_INLINE=0): two such steps in
    front of the last reader of an odd list put the first of them three steps behind the list's padding re-run."""
    _, _, exe = model
    for seed, tips in ((11, 12), (12, 6), (13, 20)):
        r = run(exe, seed, 3000, tips, 1, in_step=0)
        assert r.returncode == 0 and "VIRT_MODEL OK" in r.stdout, (seed, tips, r.stdout[-300:], r.stderr[-600:])
        assert int(r.stdout.split(",")[-1].split()[0]) > 100, r.stdout  # odd lists did occur


def _variant(model, name, edit):
    """the bookkeeping functions of phyhip_queue.hip with `edit` applied, linked in front of the library"""
    d, obj, _ = model
    src = open(os.path.join(CSRC, "phyhip_queue.hip")).read()
    end = src.index("\n// Host-computed matrices queued")
    head = src[:end].replace('#include "phyhip_host.hpp"', '#include "%s"' % os.path.join(CSRC, "phyhip_host.hpp")) + "\n}\n"
    text = edit(head)
    p = os.path.join(d, name + ".hip")
    open(p, "w").write(text)
    o = os.path.join(d, name + ".o")
    compile_host(p, o, d)
    exe = os.path.join(d, "vm_" + name)
    link([obj, o], exe, d)
    return exe


def test_the_model_sees_the_unstored_child_of_a_padded_list(model):
    """Control: the rewriting WITHOUT the statement that stores the first of two re-issued children of an odd list's last reader."""
    def edit(head):
        fix = "      d1.pad &= ~kOpNoStore;\n"
        assert fix in head
        return head.replace(fix, "")
    exe = _variant(model, "pad_defect", edit)
    runs = [run(exe, seed, 3000, 12, 1, in_step=0) for seed in (11, 12, 13)]
    assert sum(r.returncode != 0 and "re-run of an odd list" in r.stderr for r in runs) >= 2, [(r.stdout[-200:], r.stderr[-200:]) for r in runs]


def test_the_model_sees_a_reader_left_virtual(model):
    """Control: devirtualise() without the keep-real request (round 5's second defect: a long queue that rewrites the buffer with a
    tip x tip operation left it virtual again under the reader that was about to copy it from memory)."""
    def edit(head):
        keep = "    I->keep_real_flag[buf] = 1;\n    I->keep_real.push_back(buf);\n"
        assert keep in head
        return head.replace(keep, "")
    exe = _variant(model, "keep_defect", edit)
    runs = [run(exe, seed, 3000, 12, 1) for seed in (1, 2, 3)]
    assert any(r.returncode != 0 and "VIRT_MODEL FAIL" in r.stderr for r in runs), [(r.stdout[-200:], r.stderr[-200:]) for r in runs]


def test_the_model_sees_a_stale_definition(model):
    """The bookkeeping functions of phyhip_queue.hip compiled WITHOUT the statement that makes a buffer real when a short launch
    rewrites it, linked in front of the library: the model must fail."""
    d, obj, _ = model
    src = open(os.path.join(CSRC, "phyhip_queue.hip")).read()
    end = src.index("\n// Host-computed matrices queued")
    head = src[:end].replace('#include "phyhip_host.hpp"', '#include "%s"' % os.path.join(CSRC, "phyhip_host.hpp")) + "\n}\n"
    healthy = "      if (I->virt[o.dest]) { I->virt[o.dest] = 0; --I->n_virtual; }\n    return;"
    assert healthy in head
    variants = {"as_built": head, "defect": head.replace(healthy, "    return;")}
    seen = {}
    for name, text in variants.items():
        p = os.path.join(d, name + ".hip")
        open(p, "w").write(text)
        o = os.path.join(d, name + ".o")
        compile_host(p, o, d)
        exe = os.path.join(d, "vm_" + name)
        link([obj, o], exe, d)
        seen[name] = [run(exe, seed, 3000, 12, 1) for seed in (1, 2, 3)]
    assert all(r.returncode == 0 and "VIRT_MODEL OK" in r.stdout for r in seen["as_built"]), seen["as_built"][0].stderr[-400:]
    assert all(r.returncode != 0 and "VIRT_MODEL FAIL" in r.stderr for r in seen["defect"]), [r.stdout[-200:] for r in seen["defect"]]
