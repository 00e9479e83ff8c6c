"""phyml_amd/csrc/phyhip_exp.hpp -- the device's exp() -- compiled by gcc for the host and held against this image's libm, the
one the reference calls (src/models.c:275, src/lk.c:712-713): the same double for every input tried, the branches for
subnormal results, out-of-range arguments, infinities and NaNs included.  (The device runs the same header: its FMA and its
plain multiply / add are IEEE operations like the host's; tests/test_gpu_cases.py::test_device_built_matrices_at_every_category_count
and tests/test_gpu_parity.py::test_device_pmatrices hold what it produces to the reference's matrices bit for bit.)"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SRC = r"""
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "%s/phyml_amd/csrc/phyhip_exp.hpp"
static uint64_t s = 88172645463325252ull;
static uint64_t rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; }
int main(int argc, char **argv)
{
  long bad = 0, n = atol(argv[1]);
  for (long i = 0; i < n; ++i)
  {
    double x; const uint64_t r = rnd(); const double u = (double)(r >> 11) / 9007199254740992.0;
    switch (i & 7)
    {
      case 0: x = -u * 800.0; break;                              /* what a transition matrix asks for, far out */
      case 1: x = -u * 30.0; break;
      case 2: x = -u; break;
      case 3: x = -ldexp(u, -(int)(r & 63)); break;               /* tiny arguments (the 1 + x branch below 2^-54) */
      case 4: x = (u - 0.5) * 1500.0; break;
      case 5: x = -700.0 - u * 60.0; break;                       /* subnormal results */
      case 6: x = (u - 0.5) * 4000.0; break;                      /* under- and overflow */
      default: { uint64_t b = r; memcpy(&x, &b, 8); } break;      /* any bit pattern: infinities, NaNs */
    }
    const double a = exp(x), b = phyhip_exp_ref(x, phyhip_exp_tab);
    uint64_t ab, bb; memcpy(&ab, &a, 8); memcpy(&bb, &b, 8);
    if (ab != bb && !(a != a && b != b)) { if (bad < 10) printf("x=%%a libm=%%a port=%%a\n", x, a, b); ++bad; }
  }
  printf("%%ld inputs, %%ld differ\n", n, bad);
  return bad != 0;
}
"""


def test_the_port_is_this_libms_exp(tmp_path):
    c = tmp_path / "t.c"
    c.write_text(SRC % ROOT)
    exe = str(tmp_path / "t")
    subprocess.check_call(["gcc", "-O2", "-mfma", "-ffp-contract=off", "-o", exe, str(c), "-lm"])
    r = subprocess.run([exe, "40000000"], stdout=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stdout
    assert "40000000 inputs, 0 differ" in r.stdout


def test_the_table_is_the_one_in_libm():
    """The 256 table words and the polynomial of the header are the bytes of __exp_data in this image's libm.so.6."""
    import re
    hdr = open(os.path.join(ROOT, "phyml_amd", "csrc", "phyhip_exp.hpp")).read()
    tab = [int(x, 16) for x in re.findall(r"(0x[0-9a-f]+)ull", hdr[hdr.index("phyhip_exp_tab[256]"):hdr.index("};")])]
    assert len(tab) == 256
    blob = b"".join(int(x).to_bytes(8, "little") for x in tab)
    lib = None
    for p in ("/lib/x86_64-linux-gnu/libm.so.6", "/usr/lib/x86_64-linux-gnu/libm.so.6", "/lib64/libm.so.6"):
        if os.path.exists(p):
            lib = open(p, "rb").read()
            break
    assert lib is not None
    at = lib.find(blob)
    assert at > 0
    import struct
    head = struct.unpack("<8d", lib[at - 0x70:at - 0x70 + 64])  # invln2N, shift, negln2hiN, negln2loN, C2 .. C5
    want = [float.fromhex(x) for x in ("0x1.71547652b82fep+7", "0x1.8p52", "-0x1.62e42fefa0000p-8", "-0x1.cf79abc9e3b3ap-47",
                                       "0x1.ffffffffffdbdp-2", "0x1.555555555543cp-3", "0x1.55555cf172b91p-5", "0x1.1111167a4d017p-7")]
    assert list(head) == want
