// phyhip_exp.hpp -- exp() on the device as the REFERENCE's libm computes it, bit for bit.
//
// PhyML's transition matrices are P = U diag(exp(lambda_k len)) U^-1 (src/models.c:257-326) and the eigen-basis dLk tables are
// exp(lambda_k len) again (src/lk.c:690-730): everything else of those computations the device already did in the reference's
// operation order, so the device's own exp -- a few ulp from glibc's -- was the ONE reason device-built matrices were not the
// reference's doubles (and the bit-exact route meant host PMat() + an upload per matrix).  This is glibc 2.35's exp (Szabolcs
// Nagy's table-driven algorithm of sysdeps/ieee754/dbl-64/e_exp.c, N = 128; the __exp_fma ifunc variant an FMA-capable x86-64
// runs): the same operations in the same order with the same contractions the shipped binary has -- read off its disassembly,
// constants and table read out of its .rodata (they are the published ones of ARM's optimized-routines;
// tests/test_exp_port.py::test_the_table_is_the_one_in_libm finds them in the image's libm.so.6) -- the subnormal-result and
// out-of-range branches included.  Pinned on the CPU against libm itself: tests/test_exp_port.py (this header compiled by gcc,
// 4 x 10^7 inputs per run, 2 x 10^8 when it was written: 0 differ); on the device through what it produces: transition matrices
// equal to the reference's own dumps and to the restatement's over the whole range of edge lengths, bit for bit
// (tests/test_gpu_parity.py::test_device_pmatrices, tests/test_gpu_cases.py::test_device_built_matrices_at_every_category_count).
// -ffp-contract=off (the build's flag) keeps the unfused operations unfused.
#pragma once
#include <stdint.h>
#if defined(__HIPCC__) || defined(__HIP__)
#define PHYHIP_EXP_FN __host__ __device__ __forceinline__
#define PHYHIP_EXP_TAB static __device__ const
#define PHYHIP_EXP_FMA(a, b, c) __builtin_fma(a, b, c)
#else
#include <math.h>
#define PHYHIP_EXP_FN static inline
#define PHYHIP_EXP_TAB static const
#define PHYHIP_EXP_FMA(a, b, c) fma(a, b, c)
#endif

// 2^(i/128) as {tail, scale bits - i << 45}: exp_data.tab of glibc's e_exp_data.c
PHYHIP_EXP_TAB uint64_t phyhip_exp_tab[256] = {
  0x0ull, 0x3ff0000000000000ull, 0x3c9b3b4f1a88bf6eull, 0x3feff63da9fb3335ull,
  0xbc7160139cd8dc5dull, 0x3fefec9a3e778061ull, 0xbc905e7a108766d1ull, 0x3fefe315e86e7f85ull,
  0x3c8cd2523567f613ull, 0x3fefd9b0d3158574ull, 0xbc8bce8023f98efaull, 0x3fefd06b29ddf6deull,
  0x3c60f74e61e6c861ull, 0x3fefc74518759bc8ull, 0x3c90a3e45b33d399ull, 0x3fefbe3ecac6f383ull,
  0x3c979aa65d837b6dull, 0x3fefb5586cf9890full, 0x3c8eb51a92fdeffcull, 0x3fefac922b7247f7ull,
  0x3c3ebe3d702f9cd1ull, 0x3fefa3ec32d3d1a2ull, 0xbc6a033489906e0bull, 0x3fef9b66affed31bull,
  0xbc9556522a2fbd0eull, 0x3fef9301d0125b51ull, 0xbc5080ef8c4eea55ull, 0x3fef8abdc06c31ccull,
  0xbc91c923b9d5f416ull, 0x3fef829aaea92de0ull, 0x3c80d3e3e95c55afull, 0x3fef7a98c8a58e51ull,
  0xbc801b15eaa59348ull, 0x3fef72b83c7d517bull, 0xbc8f1ff055de323dull, 0x3fef6af9388c8deaull,
  0x3c8b898c3f1353bfull, 0x3fef635beb6fcb75ull, 0xbc96d99c7611eb26ull, 0x3fef5be084045cd4ull,
  0x3c9aecf73e3a2f60ull, 0x3fef54873168b9aaull, 0xbc8fe782cb86389dull, 0x3fef4d5022fcd91dull,
  0x3c8a6f4144a6c38dull, 0x3fef463b88628cd6ull, 0x3c807a05b0e4047dull, 0x3fef3f49917ddc96ull,
  0x3c968efde3a8a894ull, 0x3fef387a6e756238ull, 0x3c875e18f274487dull, 0x3fef31ce4fb2a63full,
  0x3c80472b981fe7f2ull, 0x3fef2b4565e27cddull, 0xbc96b87b3f71085eull, 0x3fef24dfe1f56381ull,
  0x3c82f7e16d09ab31ull, 0x3fef1e9df51fdee1ull, 0xbc3d219b1a6fbffaull, 0x3fef187fd0dad990ull,
  0x3c8b3782720c0ab4ull, 0x3fef1285a6e4030bull, 0x3c6e149289cecb8full, 0x3fef0cafa93e2f56ull,
  0x3c834d754db0abb6ull, 0x3fef06fe0a31b715ull, 0x3c864201e2ac744cull, 0x3fef0170fc4cd831ull,
  0x3c8fdd395dd3f84aull, 0x3feefc08b26416ffull, 0xbc86a3803b8e5b04ull, 0x3feef6c55f929ff1ull,
  0xbc924aedcc4b5068ull, 0x3feef1a7373aa9cbull, 0xbc9907f81b512d8eull, 0x3feeecae6d05d866ull,
  0xbc71d1e83e9436d2ull, 0x3feee7db34e59ff7ull, 0xbc991919b3ce1b15ull, 0x3feee32dc313a8e5ull,
  0x3c859f48a72a4c6dull, 0x3feedea64c123422ull, 0xbc9312607a28698aull, 0x3feeda4504ac801cull,
  0xbc58a78f4817895bull, 0x3feed60a21f72e2aull, 0xbc7c2c9b67499a1bull, 0x3feed1f5d950a897ull,
  0x3c4363ed60c2ac11ull, 0x3feece086061892dull, 0x3c9666093b0664efull, 0x3feeca41ed1d0057ull,
  0x3c6ecce1daa10379ull, 0x3feec6a2b5c13cd0ull, 0x3c93ff8e3f0f1230ull, 0x3feec32af0d7d3deull,
  0x3c7690cebb7aafb0ull, 0x3feebfdad5362a27ull, 0x3c931dbdeb54e077ull, 0x3feebcb299fddd0dull,
  0xbc8f94340071a38eull, 0x3feeb9b2769d2ca7ull, 0xbc87deccdc93a349ull, 0x3feeb6daa2cf6642ull,
  0xbc78dec6bd0f385full, 0x3feeb42b569d4f82ull, 0xbc861246ec7b5cf6ull, 0x3feeb1a4ca5d920full,
  0x3c93350518fdd78eull, 0x3feeaf4736b527daull, 0x3c7b98b72f8a9b05ull, 0x3feead12d497c7fdull,
  0x3c9063e1e21c5409ull, 0x3feeab07dd485429ull, 0x3c34c7855019c6eaull, 0x3feea9268a5946b7ull,
  0x3c9432e62b64c035ull, 0x3feea76f15ad2148ull, 0xbc8ce44a6199769full, 0x3feea5e1b976dc09ull,
  0xbc8c33c53bef4da8ull, 0x3feea47eb03a5585ull, 0xbc845378892be9aeull, 0x3feea34634ccc320ull,
  0xbc93cedd78565858ull, 0x3feea23882552225ull, 0x3c5710aa807e1964ull, 0x3feea155d44ca973ull,
  0xbc93b3efbf5e2228ull, 0x3feea09e667f3bcdull, 0xbc6a12ad8734b982ull, 0x3feea012750bdabfull,
  0xbc6367efb86da9eeull, 0x3fee9fb23c651a2full, 0xbc80dc3d54e08851ull, 0x3fee9f7df9519484ull,
  0xbc781f647e5a3ecfull, 0x3fee9f75e8ec5f74ull, 0xbc86ee4ac08b7db0ull, 0x3fee9f9a48a58174ull,
  0xbc8619321e55e68aull, 0x3fee9feb564267c9ull, 0x3c909ccb5e09d4d3ull, 0x3feea0694fde5d3full,
  0xbc7b32dcb94da51dull, 0x3feea11473eb0187ull, 0x3c94ecfd5467c06bull, 0x3feea1ed0130c132ull,
  0x3c65ebe1abd66c55ull, 0x3feea2f336cf4e62ull, 0xbc88a1c52fb3cf42ull, 0x3feea427543e1a12ull,
  0xbc9369b6f13b3734ull, 0x3feea589994cce13ull, 0xbc805e843a19ff1eull, 0x3feea71a4623c7adull,
  0xbc94d450d872576eull, 0x3feea8d99b4492edull, 0x3c90ad675b0e8a00ull, 0x3feeaac7d98a6699ull,
  0x3c8db72fc1f0eab4ull, 0x3feeace5422aa0dbull, 0xbc65b6609cc5e7ffull, 0x3feeaf3216b5448cull,
  0x3c7bf68359f35f44ull, 0x3feeb1ae99157736ull, 0xbc93091fa71e3d83ull, 0x3feeb45b0b91ffc6ull,
  0xbc5da9b88b6c1e29ull, 0x3feeb737b0cdc5e5ull, 0xbc6c23f97c90b959ull, 0x3feeba44cbc8520full,
  0xbc92434322f4f9aaull, 0x3feebd829fde4e50ull, 0xbc85ca6cd7668e4bull, 0x3feec0f170ca07baull,
  0x3c71affc2b91ce27ull, 0x3feec49182a3f090ull, 0x3c6dd235e10a73bbull, 0x3feec86319e32323ull,
  0xbc87c50422622263ull, 0x3feecc667b5de565ull, 0x3c8b1c86e3e231d5ull, 0x3feed09bec4a2d33ull,
  0xbc91bbd1d3bcbb15ull, 0x3feed503b23e255dull, 0x3c90cc319cee31d2ull, 0x3feed99e1330b358ull,
  0x3c8469846e735ab3ull, 0x3feede6b5579fdbfull, 0xbc82dfcd978e9db4ull, 0x3feee36bbfd3f37aull,
  0x3c8c1a7792cb3387ull, 0x3feee89f995ad3adull, 0xbc907b8f4ad1d9faull, 0x3feeee07298db666ull,
  0xbc55c3d956dcaebaull, 0x3feef3a2b84f15fbull, 0xbc90a40e3da6f640ull, 0x3feef9728de5593aull,
  0xbc68d6f438ad9334ull, 0x3feeff76f2fb5e47ull, 0xbc91eee26b588a35ull, 0x3fef05b030a1064aull,
  0x3c74ffd70a5fddcdull, 0x3fef0c1e904bc1d2ull, 0xbc91bdfbfa9298acull, 0x3fef12c25bd71e09ull,
  0x3c736eae30af0cb3ull, 0x3fef199bdd85529cull, 0x3c8ee3325c9ffd94ull, 0x3fef20ab5fffd07aull,
  0x3c84e08fd10959acull, 0x3fef27f12e57d14bull, 0x3c63cdaf384e1a67ull, 0x3fef2f6d9406e7b5ull,
  0x3c676b2c6c921968ull, 0x3fef3720dcef9069ull, 0xbc808a1883ccb5d2ull, 0x3fef3f0b555dc3faull,
  0xbc8fad5d3ffffa6full, 0x3fef472d4a07897cull, 0xbc900dae3875a949ull, 0x3fef4f87080d89f2ull,
  0x3c74a385a63d07a7ull, 0x3fef5818dcfba487ull, 0xbc82919e2040220full, 0x3fef60e316c98398ull,
  0x3c8e5a50d5c192acull, 0x3fef69e603db3285ull, 0x3c843a59ac016b4bull, 0x3fef7321f301b460ull,
  0xbc82d52107b43e1full, 0x3fef7c97337b9b5full, 0xbc892ab93b470dc9ull, 0x3fef864614f5a129ull,
  0x3c74b604603a88d3ull, 0x3fef902ee78b3ff6ull, 0x3c83c5ec519d7271ull, 0x3fef9a51fbc74c83ull,
  0xbc8ff7128fd391f0ull, 0x3fefa4afa2a490daull, 0xbc8dae98e223747dull, 0x3fefaf482d8e67f1ull,
  0x3c8ec3bc41aa2008ull, 0x3fefba1bee615a27ull, 0x3c842b94c3a9eb32ull, 0x3fefc52b376bba97ull,
  0x3c8a64a931d185eeull, 0x3fefd0765b6e4540ull, 0xbc8e37bae43be3edull, 0x3fefdbfdad9cbe14ull,
  0x3c77893b4d91cd9dull, 0x3fefe7c1819e90d8ull, 0x3c5305c14160cc89ull, 0x3feff3c22b8f71f1ull};

PHYHIP_EXP_FN double phyhip_exp_asdouble(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
PHYHIP_EXP_FN uint64_t phyhip_exp_asuint(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

// T: the table (phyhip_exp_tab, or a copy of it nearer to the lanes: on the device any pointer type, so that a copy in LDS is
// read with LDS instructions).
// Written without divergent branches: the common path is computed by every lane, the rare results (|x| < 2^-54, |x| >= 512,
// infinities, NaN) are selected over it -- the subnormal-result arithmetic behind ONE wave-uniform test -- because nested divergent
// branches cost the two-operation short-launch kernels the scalar registers of their execution-mask saves (they reserved scratch:
// tests/test_kernel_resources.py).  Every selected value is the value the branch of the original computes.
#if defined(__HIP_DEVICE_COMPILE__)
#define PHYHIP_EXP_ANY(c) (__builtin_amdgcn_ballot_w64(c) != 0)
#else
#define PHYHIP_EXP_ANY(c) (c)
#endif
#if defined(__HIPCC__) || defined(__HIP__)
template <typename TabPtr>
PHYHIP_EXP_FN double phyhip_exp_ref(double x, TabPtr T)
#else
PHYHIP_EXP_FN double phyhip_exp_ref(double x, const uint64_t *T)
#endif
{
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52, NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47,
               C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5, C5 = 0x1.1111167a4d017p-7;
  const uint64_t xb = phyhip_exp_asuint(x);
  const uint32_t abstop = (uint32_t)(xb >> 52) & 0x7ff;
  const int      tiny = abstop < 0x3c9u;   // |x| < 2^-54: 1 + x
  const int      far = abstop > 0x408u;    // |x| >= 1024, infinities, NaN
  const int      edge = abstop == 0x408u;  // 512 <= |x| < 1024: the scale may not be representable
  // (lanes whose result is selected below compute on a harmless argument)
  const double xs = (tiny || far) ? 0.0 : x;
  double kd = PHYHIP_EXP_FMA(xs, InvLn2N, Shift);
  const uint64_t ki = phyhip_exp_asuint(kd);
  kd -= Shift;
  double r = PHYHIP_EXP_FMA(kd, NegLn2hiN, xs);
  r = PHYHIP_EXP_FMA(kd, NegLn2loN, r);
  const uint64_t idx = 2 * (ki & 127), top = ki << 45;
  const double   tail = phyhip_exp_asdouble(T[idx]);
  const uint64_t sbits = T[idx + 1] + top;
  const double   p23 = PHYHIP_EXP_FMA(C3, r, C2), tr = r + tail, r2 = r * r, p45 = PHYHIP_EXP_FMA(r, C5, C4);
  double         tmp = PHYHIP_EXP_FMA(p23, r2, tr);
  tmp = PHYHIP_EXP_FMA(r2 * r2, p45, tmp);
  const double scale = phyhip_exp_asdouble(edge ? 0x3ff0000000000000ull : sbits);
  double       res = PHYHIP_EXP_FMA(scale, tmp, scale);
  if (PHYHIP_EXP_ANY(edge))
  { // specialcase() of the original, both signs of k side by side
    const double sp = phyhip_exp_asdouble(sbits - (1009ull << 52));
    const double yp = 0x1p1009 * PHYHIP_EXP_FMA(sp, tmp, sp);          // k > 0: the exponent of scale might have overflowed
    const double sn = phyhip_exp_asdouble(sbits + (1022ull << 52));    // k < 0: the result may be subnormal
    const double st = tmp * sn;
    double       y = sn + st;
    {
      const double hi = y + 1.0;
      double       lo = sn - y;
      lo = lo + st;
      double t = 1.0 - hi;
      t = t + y;
      t = t + lo;
      t = t + hi;
      double y2 = t - 1.0;
      y2 = (y2 == 0.0) ? 0.0 : y2;
      y = (y < 1.0) ? y2 : y;
    }
    const double yn = 0x1p-1022 * y;
    const double e = (ki & 0x80000000ull) ? yn : yp;
    res = edge ? e : res;
  }
  // |x| >= 1024: -inf -> 0, NaN / +inf -> 1 + x, negative -> 0 (underflow), positive -> +inf (overflow)
  const double f = (xb == 0xfff0000000000000ull) ? 0.0 : (abstop == 0x7ffu) ? 1.0 + x : (xb >> 63) ? 0.0 : phyhip_exp_asdouble(0x7ff0000000000000ull);
  res = far ? f : res;
  res = tiny ? 1.0 + x : res;
  return res;
}
