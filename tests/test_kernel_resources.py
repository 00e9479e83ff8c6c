"""Compiler-dependent properties of the hot kernels, pinned on the code objects inside the BUILT product library
(phyml_amd/lib/libphyhip.so -- the file that travels to the GPU box): no scratch, and register counts inside the occupancy each
design assumes.  A ROCm point release that starts spilling would otherwise turn e.g. the 24 us resident SPR candidate into a
scratch-bound one without any test noticing.  CPU-only: reads the AMDGPU metadata notes with the ROCm LLVM tools."""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "llvm", "bin")
LIB = os.path.join(ROOT, "phyml_amd", "lib", "libphyhip.so")

pytestmark = pytest.mark.skipif(not (os.path.exists(os.path.join(LLVM, "llvm-objdump")) and os.path.exists(LIB)),
                                reason="needs the ROCm LLVM tools and the built library")


def kernels_of(path, tmp):
    """{mangled name: {field: int}} of every kernel in the gfx950 code objects bundled into `path`"""
    work = os.path.join(tmp, "x")
    os.makedirs(work, exist_ok=True)
    local = os.path.join(work, os.path.basename(path))
    shutil.copy(path, local)
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", local], cwd=work, check=True, stdout=subprocess.DEVNULL,
                   stderr=subprocess.DEVNULL)
    out = {}
    for f in sorted(os.listdir(work)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(work, f)], check=True,
                               stdout=subprocess.PIPE, text=True).stdout
        for blk in notes.split("  - .agpr_count:")[1:]:
            blk = ".agpr_count:" + blk
            name = re.search(r"\.name:\s*(\S+)", blk).group(1)
            out[name] = {k: int(v) for k, v in re.findall(r"\.(\w+):\s*(\d+)\s*$", blk, flags=re.M)}
    return out


@pytest.fixture(scope="module")
def product(tmp_path_factory):
    k = kernels_of(LIB, str(tmp_path_factory.mktemp("kres")))
    assert len(k) > 80, len(k)
    return k


def waves_per_simd(vgprs):
    """gfx950: 512 unified registers per SIMD lane, allocated in blocks of 8"""
    return 512 // (((vgprs + 7) // 8) * 8)


def test_no_hot_kernel_uses_scratch(product):
    hot = [n for n in product if re.search(r"traverse_nt2_kernel|traverse_aa_kernel|resident_big_kernel|resident_nt2_kernel|"
                                           r"resident_dlk_kernel|dlk64_kernel|dlk_kernel|eigen_lr_kernel|pmat_kernel|pmat20_kernel", n)]
    assert len(hot) > 60
    # (the one-operation-ahead form of the lane-per-pattern kernel, DIST = 1, is selected by a diag-build switch only -- the product
    # never launches it; its <4 categories, 4 lane groups> shape spills 48 bytes under a launch bound of five waves per SIMD)
    hot = [n for n in hot if not re.search(r"traverse_nt2_kernelILi\dELi\dELb0ELi0ELi1ELb0EEEv", n)]
    for n in hot:
        k = product[n]
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (n, k)


def test_default_traversal_kernels_keep_their_occupancy(product):
    # traverse_nt2_kernel<C, G, DBG=false, ARGS=0, DIST=2>: G waves per SIMD is what __launch_bounds__(64, G) asks for and what
    # the choice of G = 2 below ~100 k patterns rests on
    # ... <..., INL>: the instantiation for lists with in-step tip x tip children (four matrices staged per step) as well
    for c, g in ((4, 2), (4, 1), (2, 2), (2, 1), (1, 1), (3, 1)):
        for inl in (0, 1):
            n = f"_ZN6phyhip19traverse_nt2_kernelILi{c}ELi{g}ELb0ELi0ELi2ELb{inl}EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdPKhPy"
            assert waves_per_simd(product[n]["vgpr_count"]) >= g, (n, product[n]["vgpr_count"])
    # the launch with two wave shapes (two-lane waves + four-lane waves for the rest): two per SIMD like the two-lane kernel
    for inl in (0, 1):
        n = f"_ZN6phyhip25traverse_nt2_mixed_kernelILi4ELb{inl}EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdPKhi"
        assert waves_per_simd(product[n]["vgpr_count"]) >= 2 and product[n]["private_segment_fixed_size"] == 0, (n, product[n])
    # the 20-state kernel: 1 loader + 15 consumer waves per workgroup = four per SIMD
    # (<C, DBG, ABL, ARGS, INL, NT, D2, RES>: list form, argument form, and the list form with in-step tip x tip children)
    for c in (1, 2, 3, 4):
        for args, inl in ((0, 0), (1, 0), (0, 1)):
            n = f"_ZN6phyhip18traverse_aa_kernelILi{c}ELb0ELi0ELb{args}ELb{inl}ELi1ELb0ELb0EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdiPKjPyNS_10AaResidentE"
            assert product[n]["vgpr_count"] <= 128, (n, product[n]["vgpr_count"])
        # the resident form (<..., RES>): 1 loader + 7 consumers = two waves per SIMD
        n = f"_ZN6phyhip18traverse_aa_kernelILi{c}ELb0ELi0ELb1ELb0ELi1ELb0ELb1EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdiPKjPyNS_10AaResidentE"
        assert product[n]["vgpr_count"] <= 256 and product[n]["private_segment_fixed_size"] == 0, (n, product[n])


def test_measured_variants_of_the_20_state_kernel_fit_their_waves(tmp_path):
    """The diag build's A/B variants of the 20-state list form (profiles/r06_aa_kernel.md): two wave-tiles per consumer wave
    (1 loader + 7 consumers = two waves per SIMD: 256 registers) and loads two operations ahead (1 + 11 = three per SIMD: 168)."""
    diag = os.path.join(ROOT, "phyml_amd", "lib_diag", "libphyhip.so")
    if not os.path.exists(diag):
        pytest.skip("no diag build")
    k = kernels_of(diag, str(tmp_path))
    for c in (1, 2, 3, 4):
        for inl in (0, 1):
            n = f"_ZN6phyhip18traverse_aa_kernelILi{c}ELb0ELi0ELb0ELb{inl}ELi2ELb0ELb0EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdiPKjPyNS_10AaResidentE"
            assert k[n]["vgpr_count"] <= 256 and k[n]["private_segment_fixed_size"] == 0, (n, k[n])
            n = f"_ZN6phyhip18traverse_aa_kernelILi{c}ELb0ELi0ELb0ELb{inl}ELi1ELb1ELb0EEEvNS_10TreeParamsEPKNS_8IssueRecEPKNS_7ExecRecEPKdiPKjPyNS_10AaResidentE"
            assert k[n]["vgpr_count"] <= 168 and k[n]["private_segment_fixed_size"] == 0, (n, k[n])


def test_large_grid_resident_kernel_fits_its_waves(product):
    """resident_big_kernel<C, G, NW>: one workgroup of NW waves per CU must be resident (NW / 4 per SIMD), and no shape spills --
    which is what `-mllvm -disable-machine-licm` is there for (phyhip_big.hip's header)."""
    shapes = [n for n in product if "resident_big_kernel" in n]
    assert len(shapes) == 6
    for n in shapes:
        nw = int(re.search(r"resident_big_kernelILi\d+ELi\d+ELi(\d+)E", n).group(1))
        k = product[n]
        assert waves_per_simd(k["vgpr_count"]) >= nw // 4, (n, k["vgpr_count"], nw)
        assert k["private_segment_fixed_size"] == 0 and k["vgpr_spill_count"] == 0, (n, k)
        assert k["group_segment_fixed_size"] <= 160 * 1024


def test_the_build_keeps_the_flag_and_its_absence_would_be_seen(tmp_path):
    """The flag is in the build recipe, and the checks above do see what happens without it: phyhip_big.hip compiled plainly
    spills in (nearly) every shape (11 s)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("graft_entry", os.path.join(ROOT, "__graft_entry__.py"))
    ge = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ge)
    flags = dict(ge.UNITS)["phyhip_big.hip"]
    assert "-disable-machine-licm" in flags
    hipcc = ge.HIPCC
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    asm = os.path.join(str(tmp_path), "big_plain.s")
    subprocess.run([hipcc] + ge.CFLAGS + ["-S", "--cuda-device-only", "-o", asm, os.path.join(ROOT, "phyml_amd", "csrc", "phyhip_big.hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=str(tmp_path))
    txt = open(asm).read()
    spills = []
    for blk in txt.split("  - .agpr_count:")[1:]:
        if "resident_big_kernel" in re.search(r"\.name:\s*(\S+)", blk).group(1):
            spills.append(int(re.search(r"\.private_segment_fixed_size:\s*(\d+)", blk).group(1)))
    # (every shape did in round 5; with the round-6 sources one of the six escapes -- the check sees the flag's absence all the same)
    assert len(spills) == 6 and sum(1 for x in spills if x > 0) >= 4, spills
