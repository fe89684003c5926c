// kernels_wave4096.hpp -- ONE WAVE per Doppler row for n = 4096 (L = 8192), complex64 (measurement build,
// CAF_ROW_KERNEL=4; VERDICT r03 item 6: one structural attempt on the complex64 row kernel).
//
// k_duo_rows<float> spends half of its time in LDS exchanges, workgroup barriers and latency that three waves per
// SIMD do not hide (0.55 of its own issue ceiling).  This kernel removes the barriers altogether and halves the LDS
// traffic: a row belongs to ONE wave, every lane holds 64 points (4096 = 64 x 64) and a 4096-point transform is
//     radix-64 over the register index  ->  lane twiddles  ->  ONE wave-local 64 x 64 transposition through LDS
//     ->  radix-64 over the register index,
// natural order in, natural order out (element m at lane m % 64, register m / 64), so the four transforms of a row
// (two chains x forward / inverse) are the same routine, the haystack spectrum is multiplied in natural order and the
// last radix-2 stage E[m] +- W_8192^m O[m] pairs registers of the same lane.  No s_barrier anywhere: LDS operations of
// one wave execute in issue order.  One wave per SIMD (512 registers: the even chain's result waits in the upper
// half of the register file while the odd chain runs), four one-wave workgroups of 33 KiB LDS per CU.
//
//   radix-64 in registers = 8 x 8: sixteen 8-point butterflies + 49 compile-time twiddles W_64^(r0 ka)
//   lane twiddles W_4096^(lane * ka), ka = 8a + b: fourteen held values A[a] = W^(8 a lane), B[b] = W^(b lane); the
//       mixer's lane factor (forward) and the last stage's W_8192^ka (inverse, odd chain) ride on them
//   LDS image: 64 rows of 65 complex (one pad): lane n1 writes register ka to row ka (16 contiguous lanes per
//       ds_write_b64 group), lane ka reads its row (bank (2 ka + {0,1}) mod 64 over a 32-lane group): conflict-free
//   per-row phasors (k_wave_phasors): lo[n1] = e^{i ph n1} (one load per lane) and the wave-uniform register steps
//       conj(e^{i ph 64 r}) (x e^{2 pi i r / 128} for the odd chain) as scalar operands of the mixer multiply
#pragma once
#include "../kernels_seq4096.hpp"

namespace caf {

constexpr int WV_ROW = 65;    // LDS row stride in elements
constexpr int WV_PH = 192;    // phasor entries per row: lo[64] | cstep0[64] | cstep1[64]
constexpr size_t wave_lds_bytes() { return (size_t)64 * WV_ROW * sizeof(cpx<float>); }

template <typename T>
__global__ void k_wave_phasors(const double *__restrict__ ph, int nrows, cpx<T> *__restrict__ tab)
{
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = g / WV_PH, e = g - row * WV_PH;
    if (row > nrows) return;
    const double p = row < nrows ? ph[row] : 0.0;   // row `nrows`: the f = 0 row of the haystack transform
    const int j = e & 63, kind = e >> 6;
    double s, c, s2 = 0.0, c2 = 1.0;
    sincos(p * ((kind == 0 ? 1.0 : 64.0) * (double)j), &s, &c);
    if (kind == 0) {
        tab[(size_t)row * WV_PH + e] = {(T)c, (T)s};   // lo[n1] = e^{i ph n1}
        return;
    }
    if (kind == 2) sincospi(2.0 * (double)j / 128.0, &s2, &c2);   // odd chain: e^{+2 pi i 64 r / 8192}
    // conj(e^{i ph 64 r}) * (c2 + i s2)
    tab[(size_t)row * WV_PH + e] = {(T)(c * c2 + s * s2), (T)(c * s2 - s * c2)};
}

// ---- v * w for a wave-uniform w held in a scalar register pair {c, s}: two packed instructions, one SGPR pair per
// distinct constant (a Linzer-Feig rotation + scale needs two) ----
__device__ __forceinline__ cpx<float> cmul_s(cpx<float> v, cpx<float> w)
{
    caf_v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(t) : "v"(pk(v)), "s"(pk(w)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]" : "=v"(r) : "v"(pk(v)), "s"(pk(w)), "v"(t));
    return unpk(r);
}
__device__ __forceinline__ cpx<double> cmul_s(cpx<double> v, cpx<double> w) { return cmul(v, w); }

// compile-time unit constant c + i s times a value
template <typename T>
__device__ __forceinline__ cpx<T> mul_k(cpx<T> v, double c, double s)
{
    if (s == 0.0) return c < 0.0 ? cpx<T>{-v.x, -v.y} : v;
    if (c == 0.0) return s < 0.0 ? cpx<T>{v.y, -v.x} : muli(v);
    return cmul_s(v, cpx<T>{(T)c, (T)s});
}

// s * conj(v) for a wave-uniform s (scalar registers): the mixer (mod.rs:46-65) on the conjugated input
__device__ __forceinline__ cpx<float> cmulc_s(cpx<float> s, cpx<float> v)
{
    caf_v2f t, r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0]" : "=v"(t) : "s"(pk(s)), "v"(pk(v)));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[0,1,1] neg_hi:[1,0,0]" : "=v"(r) : "s"(pk(s)), "v"(pk(v)), "v"(t));
    return unpk(r);
}
__device__ __forceinline__ cpx<double> cmulc_s(cpx<double> s, cpx<double> v) { return cmulc(s, v); }

// compile-time twiddle constants (indices are constants after unrolling)
// e^{2 pi i k / 64}, k < 64
__device__ constexpr double WV64_C[64] = {1.0, 0.99518472667219688624483695310947992, 0.98078528040323044912618223613423904, 0.95694033573220886493579788698026997, 0.92387953251128675612818318939678829, 0.88192126434835502971275686366038835, 0.83146961230254523707878837761790576, 0.7730104533627369608109066097584698, 0.70710678118654752440084436210484904, 0.63439328416364549821517161322549337, 0.55557023301960222474283081394853287, 0.47139673682599764855638762590525438, 0.38268343236508977172845998403039887, 0.29028467725446236763619237581739527, 0.19509032201612826784828486847702224, 0.098017140329560601994195563888641846, 0.0, -0.098017140329560601994195563888641846, -0.19509032201612826784828486847702224, -0.29028467725446236763619237581739527, -0.38268343236508977172845998403039887, -0.47139673682599764855638762590525438, -0.55557023301960222474283081394853287, -0.63439328416364549821517161322549337, -0.70710678118654752440084436210484904, -0.7730104533627369608109066097584698, -0.83146961230254523707878837761790576, -0.88192126434835502971275686366038835, -0.92387953251128675612818318939678829, -0.95694033573220886493579788698026997, -0.98078528040323044912618223613423904, -0.99518472667219688624483695310947992, -1.0, -0.99518472667219688624483695310947992, -0.98078528040323044912618223613423904, -0.95694033573220886493579788698026997, -0.92387953251128675612818318939678829, -0.88192126434835502971275686366038835, -0.83146961230254523707878837761790576, -0.7730104533627369608109066097584698, -0.70710678118654752440084436210484904, -0.63439328416364549821517161322549337, -0.55557023301960222474283081394853287, -0.47139673682599764855638762590525438, -0.38268343236508977172845998403039887, -0.29028467725446236763619237581739527, -0.19509032201612826784828486847702224, -0.098017140329560601994195563888641846, 0.0, 0.098017140329560601994195563888641846, 0.19509032201612826784828486847702224, 0.29028467725446236763619237581739527, 0.38268343236508977172845998403039887, 0.47139673682599764855638762590525438, 0.55557023301960222474283081394853287, 0.63439328416364549821517161322549337, 0.70710678118654752440084436210484904, 0.7730104533627369608109066097584698, 0.83146961230254523707878837761790576, 0.88192126434835502971275686366038835, 0.92387953251128675612818318939678829, 0.95694033573220886493579788698026997, 0.98078528040323044912618223613423904, 0.99518472667219688624483695310947992};
__device__ constexpr double WV64_S[64] = {0.0, 0.098017140329560601994195563888641846, 0.19509032201612826784828486847702224, 0.29028467725446236763619237581739527, 0.38268343236508977172845998403039887, 0.47139673682599764855638762590525438, 0.55557023301960222474283081394853287, 0.63439328416364549821517161322549337, 0.70710678118654752440084436210484904, 0.7730104533627369608109066097584698, 0.83146961230254523707878837761790576, 0.88192126434835502971275686366038835, 0.92387953251128675612818318939678829, 0.95694033573220886493579788698026997, 0.98078528040323044912618223613423904, 0.99518472667219688624483695310947992, 1.0, 0.99518472667219688624483695310947992, 0.98078528040323044912618223613423904, 0.95694033573220886493579788698026997, 0.92387953251128675612818318939678829, 0.88192126434835502971275686366038835, 0.83146961230254523707878837761790576, 0.7730104533627369608109066097584698, 0.70710678118654752440084436210484904, 0.63439328416364549821517161322549337, 0.55557023301960222474283081394853287, 0.47139673682599764855638762590525438, 0.38268343236508977172845998403039887, 0.29028467725446236763619237581739527, 0.19509032201612826784828486847702224, 0.098017140329560601994195563888641846, 0.0, -0.098017140329560601994195563888641846, -0.19509032201612826784828486847702224, -0.29028467725446236763619237581739527, -0.38268343236508977172845998403039887, -0.47139673682599764855638762590525438, -0.55557023301960222474283081394853287, -0.63439328416364549821517161322549337, -0.70710678118654752440084436210484904, -0.7730104533627369608109066097584698, -0.83146961230254523707878837761790576, -0.88192126434835502971275686366038835, -0.92387953251128675612818318939678829, -0.95694033573220886493579788698026997, -0.98078528040323044912618223613423904, -0.99518472667219688624483695310947992, -1.0, -0.99518472667219688624483695310947992, -0.98078528040323044912618223613423904, -0.95694033573220886493579788698026997, -0.92387953251128675612818318939678829, -0.88192126434835502971275686366038835, -0.83146961230254523707878837761790576, -0.7730104533627369608109066097584698, -0.70710678118654752440084436210484904, -0.63439328416364549821517161322549337, -0.55557023301960222474283081394853287, -0.47139673682599764855638762590525438, -0.38268343236508977172845998403039887, -0.29028467725446236763619237581739527, -0.19509032201612826784828486847702224, -0.098017140329560601994195563888641846};
// e^{2 pi i k / 128}, k < 64
__device__ constexpr double WV128_C[64] = {1.0, 0.99879545620517239271477160475910069, 0.99518472667219688624483695310947992, 0.98917650996478097345167373801624306, 0.98078528040323044912618223613423904, 0.97003125319454399260398420728610025, 0.95694033573220886493579788698026997, 0.94154406518302077841250940259950236, 0.92387953251128675612818318939678829, 0.90398929312344333158620029723053705, 0.88192126434835502971275686366038835, 0.85772861000027206990226998428477014, 0.83146961230254523707878837761790576, 0.80320753148064490980667651296314192, 0.7730104533627369608109066097584698, 0.74095112535495909117561689749516273, 0.70710678118654752440084436210484904, 0.6715589548470184006253768504274218, 0.63439328416364549821517161322549337, 0.59569930449243334346703652882996989, 0.55557023301960222474283081394853287, 0.51410274419322172659369383896881577, 0.47139673682599764855638762590525438, 0.42755509343028209432096685688879853, 0.38268343236508977172845998403039887, 0.33688985339222005068925321261914757, 0.29028467725446236763619237581739527, 0.24298017990326388994827416207747112, 0.19509032201612826784828486847702224, 0.14673047445536175165885012964671782, 0.098017140329560601994195563888641846, 0.049067674327418014254954976942682658, 0.0, -0.049067674327418014254954976942682658, -0.098017140329560601994195563888641846, -0.14673047445536175165885012964671782, -0.19509032201612826784828486847702224, -0.24298017990326388994827416207747112, -0.29028467725446236763619237581739527, -0.33688985339222005068925321261914757, -0.38268343236508977172845998403039887, -0.42755509343028209432096685688879853, -0.47139673682599764855638762590525438, -0.51410274419322172659369383896881577, -0.55557023301960222474283081394853287, -0.59569930449243334346703652882996989, -0.63439328416364549821517161322549337, -0.6715589548470184006253768504274218, -0.70710678118654752440084436210484904, -0.74095112535495909117561689749516273, -0.7730104533627369608109066097584698, -0.80320753148064490980667651296314192, -0.83146961230254523707878837761790576, -0.85772861000027206990226998428477014, -0.88192126434835502971275686366038835, -0.90398929312344333158620029723053705, -0.92387953251128675612818318939678829, -0.94154406518302077841250940259950236, -0.95694033573220886493579788698026997, -0.97003125319454399260398420728610025, -0.98078528040323044912618223613423904, -0.98917650996478097345167373801624306, -0.99518472667219688624483695310947992, -0.99879545620517239271477160475910069};
__device__ constexpr double WV128_S[64] = {0.0, 0.049067674327418014254954976942682658, 0.098017140329560601994195563888641846, 0.14673047445536175165885012964671782, 0.19509032201612826784828486847702224, 0.24298017990326388994827416207747112, 0.29028467725446236763619237581739527, 0.33688985339222005068925321261914757, 0.38268343236508977172845998403039887, 0.42755509343028209432096685688879853, 0.47139673682599764855638762590525438, 0.51410274419322172659369383896881577, 0.55557023301960222474283081394853287, 0.59569930449243334346703652882996989, 0.63439328416364549821517161322549337, 0.6715589548470184006253768504274218, 0.70710678118654752440084436210484904, 0.74095112535495909117561689749516273, 0.7730104533627369608109066097584698, 0.80320753148064490980667651296314192, 0.83146961230254523707878837761790576, 0.85772861000027206990226998428477014, 0.88192126434835502971275686366038835, 0.90398929312344333158620029723053705, 0.92387953251128675612818318939678829, 0.94154406518302077841250940259950236, 0.95694033573220886493579788698026997, 0.97003125319454399260398420728610025, 0.98078528040323044912618223613423904, 0.98917650996478097345167373801624306, 0.99518472667219688624483695310947992, 0.99879545620517239271477160475910069, 1.0, 0.99879545620517239271477160475910069, 0.99518472667219688624483695310947992, 0.98917650996478097345167373801624306, 0.98078528040323044912618223613423904, 0.97003125319454399260398420728610025, 0.95694033573220886493579788698026997, 0.94154406518302077841250940259950236, 0.92387953251128675612818318939678829, 0.90398929312344333158620029723053705, 0.88192126434835502971275686366038835, 0.85772861000027206990226998428477014, 0.83146961230254523707878837761790576, 0.80320753148064490980667651296314192, 0.7730104533627369608109066097584698, 0.74095112535495909117561689749516273, 0.70710678118654752440084436210484904, 0.6715589548470184006253768504274218, 0.63439328416364549821517161322549337, 0.59569930449243334346703652882996989, 0.55557023301960222474283081394853287, 0.51410274419322172659369383896881577, 0.47139673682599764855638762590525438, 0.42755509343028209432096685688879853, 0.38268343236508977172845998403039887, 0.33688985339222005068925321261914757, 0.29028467725446236763619237581739527, 0.24298017990326388994827416207747112, 0.19509032201612826784828486847702224, 0.14673047445536175165885012964671782, 0.098017140329560601994195563888641846, 0.049067674327418014254954976942682658};
// e^{2 pi i k / 8192}, k < 8
__device__ constexpr double WV8192_C[8] = {1.0, 0.99999970586288221916022821773876568, 0.99999882345170190992902571017152602, 0.99999735276697817206893996965636662, 0.9999952938095761715115801257001199, 0.99999264658070713984866211790699616, 0.99998941108192837361947235727373284, 0.99998558731514323339475029499803202};
__device__ constexpr double WV8192_S[8] = {0.0, 0.00076699031874270452693856835794857664, 0.0015339801862847656123036971502640791, 0.0023009691514258052442355523408672542, 0.0030679567629659762701453654909198425, 0.0038349425697062278259606029946348064, 0.0046019261204485707649016992969119674, 0.005368906963996343085634209182070248};
// e^{2 pi i k / 1024}, k < 8
__device__ constexpr double WV1024_C[8] = {1.0, 0.99998117528260114265699043772856772, 0.99992470183914454092164649119638322, 0.99983058179582342201572227492266551, 0.9996988186962042201157656496661722, 0.99952941750109316307970332215674097, 0.99932238458834950089622101113991035, 0.99907772775264538288878199686412614};
__device__ constexpr double WV1024_S[8] = {0.0, 0.0061358846491544753596402345903725809, 0.012271538285719926079408261951003212, 0.018406729905804820927366313014840127, 0.024541228522912288031734529459282925, 0.030674803176636625934021027565223713, 0.036807222941358832324332690927951301, 0.042938256934940823077124540281783955};

// 8-point butterfly, positive exponent, natural order in and out (26 packed instructions in complex64)
template <typename T>
__device__ __forceinline__ void wdft8(cpx<T> &x0, cpx<T> &x1, cpx<T> &x2, cpx<T> &x3, cpx<T> &x4, cpx<T> &x5, cpx<T> &x6,
                                      cpx<T> &x7)
{
    constexpr double R = 0.70710678118654752440084436210485;
    dft4(x0, x2, x4, x6);   // E[0..3]
    dft4(x1, x3, x5, x7);   // O[0..3]
    cpx<T> X1, X5, X3, X7;
    const cpx<T> X0 = x0 + x1, X4 = x0 - x1;          // E0 +- O0
    bfly_w(x2, x3, R, R, X1, X5);                     // E1 +- W8 O1
    const cpx<T> X2 = add_i(x4, x5), X6 = sub_i(x4, x5);   // E2 +- i O2
    bfly_w(x6, x7, -R, R, X3, X7);                    // E3 +- W8^3 O3
    x0 = X0; x1 = X1; x2 = X2; x3 = X3; x4 = X4; x5 = X5; x6 = X6; x7 = X7;
}

// 64-point transform over the register index, positive exponent, natural order in and out
template <typename T>
__device__ __forceinline__ void wdft64(cpx<T> (&v)[64])
{
#pragma unroll
    for (int r0 = 0; r0 < 8; ++r0)   // over r1 of v[r0 + 8 r1] -> t[r0][ka] at v[r0 + 8 ka]
        wdft8(v[r0], v[r0 + 8], v[r0 + 16], v[r0 + 24], v[r0 + 32], v[r0 + 40], v[r0 + 48], v[r0 + 56]);
#pragma unroll
    for (int ka = 1; ka < 8; ++ka)
#pragma unroll
        for (int r0 = 1; r0 < 8; ++r0) v[r0 + 8 * ka] = mul_k(v[r0 + 8 * ka], WV64_C[r0 * ka], WV64_S[r0 * ka]);
#pragma unroll
    for (int ka = 0; ka < 8; ++ka)   // over r0 of v[8 ka + r0] -> X[ka + 8 kb] at v[8 ka + kb]
        wdft8(v[8 * ka], v[8 * ka + 1], v[8 * ka + 2], v[8 * ka + 3], v[8 * ka + 4], v[8 * ka + 5], v[8 * ka + 6], v[8 * ka + 7]);
#pragma unroll
    for (int i = 0; i < 8; ++i)      // 8 x 8 transposition of the register NAMES: X[k] at v[k]
#pragma unroll
        for (int j = i + 1; j < 8; ++j) swp(v[8 * i + j], v[8 * j + i]);
}

template <typename T>
struct WvLane {
    cpx<T> A[8];   // A[a] = W_4096^(8 a lane)   (A[0] unused)
    cpx<T> B[8];   // B[b] = W_4096^(b lane)     (B[0] unused)
    cpx<T> th;     // e^{2 pi i lane / 8192}
    int lane;
};

// lane twiddles (times the lane factors folded into Bx[0..7], Ax[1..7]) + the 64 x 64 transposition through LDS
template <typename T, bool B0_IS_ONE>
__device__ __forceinline__ void wave_twiddle_transpose(cpx<T> (&v)[64], const cpx<T> (&Ax)[8], const cpx<T> (&Bx)[8],
                                                       cpx<T> *__restrict__ Lx, int lane)
{
    cpx<T> *const wr = Lx + lane;            // row ka, column lane
#pragma unroll
    for (int ka = 0; ka < 64; ++ka) {
        const int a = ka >> 3, b = ka & 7;
        cpx<T> x = v[ka];
        if (b != 0 || !B0_IS_ONE) x = cmul(x, Bx[b]);
        if (a != 0) x = cmul(x, Ax[a]);
        wr[ka * WV_ROW] = x;
    }
    wave_lds_fence();
    const cpx<T> *const rd = Lx + lane * WV_ROW;   // this lane's row
#pragma unroll
    for (int r0 = 0; r0 < 8; ++r0)        // in the order the next butterfly stage consumes them (r0 + 8 r1)
#pragma unroll
        for (int r1 = 0; r1 < 8; ++r1) v[r0 + 8 * r1] = rd[r0 + 8 * r1];
    wave_lds_fence();
}

// One chain of one row.  in: a[r] = needle[lane + 64 r]; out: v[r] = y_CH[lane + 64 r] (the odd chain's output carries
// W_8192^lane of the last radix-2 stage).
template <typename T, int CH>
__device__ __forceinline__ void wave_chain(cpx<T> (&v)[64], const cpx<T> (&a)[64], const cpx<T> *__restrict__ cstep, const cpx<T> f,
                                           const WvLane<T> &W, cpx<T> *__restrict__ Lx, const __amdgpu_buffer_rsrc_t rs_spec)
{
    using C = cpx<T>;
    // ---- mixer (mod.rs:46-65) on the conjugated needle; the lane factor f rides on the twiddles below ----
#pragma unroll
    for (int r = 0; r < 64; ++r) v[r] = cmulc_s(cstep[r], a[r]);
    wdft64(v);
    {
        C Bf[8];
        Bf[0] = f;
#pragma unroll
        for (int b = 1; b < 8; ++b) Bf[b] = cmul(W.B[b], f);
        wave_twiddle_transpose<T, false>(v, W.A, Bf, Lx, W.lane);
    }
    // haystack spectrum of this chain, natural order (issued before the second butterfly: L2 latency under it)
    C h[64];
    const unsigned voff = (unsigned)((CH * 4096 + W.lane) * sizeof(C));
#pragma unroll
    for (int k = 0; k < 64; ++k) h[k] = bload(rs_spec, voff, (unsigned)(64 * k * sizeof(C)), (C *)nullptr);
    wdft64(v);
    // ---- spectrum product (xcor_rustfft.rs:64-73) ----
#pragma unroll
    for (int k = 0; k < 64; ++k) v[k] = cmul(v[k], h[k]);
    // ---- inverse ----
    wdft64(v);
    if constexpr (CH == 0) {
        wave_twiddle_transpose<T, true>(v, W.A, W.B, Lx, W.lane);
    } else {   // (W_8192^(2 lane + 1))^ka = W_4096^(lane ka) * W_8192^ka: the last stage's lane factor, folded
        C Ai[8], Bi[8];
        Ai[0] = W.A[0];
        Bi[0] = W.B[0];
#pragma unroll
        for (int j = 1; j < 8; ++j) {
            Ai[j] = mul_k(W.A[j], WV1024_C[j], WV1024_S[j]);
            Bi[j] = mul_k(W.B[j], WV8192_C[j], WV8192_S[j]);
        }
        wave_twiddle_transpose<T, true>(v, Ai, Bi, Lx, W.lane);
    }
    wdft64(v);
}

template <typename T>
__device__ __forceinline__ void wave_lane_init(WvLane<T> &W, const FusedTables<T> &tab)
{
    W.lane = threadIdx.x & 63;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        W.A[j] = tab.tw4096[(8 * j * W.lane) & 4095];
        W.B[j] = tab.tw4096[j * W.lane];
    }
    W.th = tab.th[W.lane];
}

// ---- haystack spectrum in natural order: spec[b][chain][k] = FFT_8192(haystack ++ 0)[2 k + chain] / 8192 ----
template <typename T, int CH>
__device__ __forceinline__ void wave_prepare_one(const cpx<T> *__restrict__ sig, cpx<T> *__restrict__ spec, const WvLane<T> &W,
                                                 cpx<T> *__restrict__ Lx)
{
    using C = cpx<T>;
    C v[64];
#pragma unroll
    for (int r = 0; r < 64; ++r) {
        const C x = conj(sig[W.lane + 64 * r]);
        v[r] = CH ? mul_k(x, WV128_C[r], WV128_S[r]) : x;   // conj(h e^{-2 pi i n / 8192}): register part
    }
    wdft64(v);
    {
        C Bf[8];
        Bf[0] = CH ? W.th : C{T(1), T(0)};
#pragma unroll
        for (int b = 1; b < 8; ++b) Bf[b] = CH ? cmul(W.B[b], W.th) : W.B[b];
        if constexpr (CH) wave_twiddle_transpose<T, false>(v, W.A, Bf, Lx, W.lane);
        else wave_twiddle_transpose<T, true>(v, W.A, Bf, Lx, W.lane);
    }
    wdft64(v);
    const T inv = T(1.0 / 8192.0);
#pragma unroll
    for (int k = 0; k < 64; ++k) spec[CH * 4096 + W.lane + 64 * k] = {v[k].x * inv, -v[k].y * inv};
}

template <typename T>
__global__ __launch_bounds__(64, 1) void k_wave_prepare(const FusedArgs<T> A)
{
    __shared__ __attribute__((aligned(16))) unsigned char smem[wave_lds_bytes()];
    cpx<T> *const Lx = reinterpret_cast<cpx<T> *>(smem);
    WvLane<T> W;
    wave_lane_init(W, A.tab);
    if (blockIdx.x == 0 && W.lane == 0 && A.work) *A.work = 0u;
    for (int w = blockIdx.x; w < 2 * A.total; w += (int)gridDim.x) {
        const int b = w >> 1;
        const cpx<T> *sig = A.sig + (size_t)b * F_N;
        cpx<T> *spec = A.spec + (size_t)b * (2 * 4096);
        if (w & 1) wave_prepare_one<T, 1>(sig, spec, W, Lx);
        else wave_prepare_one<T, 0>(sig, spec, W, Lx);
    }
}

// AUX: cache policy of the surface stores (CAF_AUX_SC1 = write-through like the workgroup kernels, 0 = default)
template <typename T, int AUX = CAF_AUX_SC1>
__global__ __launch_bounds__(64, 1) void k_wave_rows(const FusedArgs<T> A, const cpx<T> *__restrict__ phasor)
{
    using C = cpx<T>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[wave_lds_bytes()];
    C *const Lx = reinterpret_cast<C *>(smem);
    WvLane<T> W;
    wave_lane_init(W, A.tab);
    const int lane = W.lane;
    const int mpair = lane & ~1;
    const bool odd = lane & 1;
    constexpr unsigned long long EVEN_LANES = 0x5555555555555555ull;

    C a[64];
    C lo_n1;
    {
        const int g0 = (int)blockIdx.x < A.total ? (int)blockIdx.x : A.total - 1;
        const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(g0 / A.rows) * F_N), 0,
                                                                             F_N * (int)sizeof(C), 0x00020000);
#pragma unroll
        for (int r = 0; r < 64; ++r) a[r] = bload(rs0, (unsigned)(lane * sizeof(C)), (unsigned)(64 * r * sizeof(C)), (C *)nullptr);
        lo_n1 = phasor[(size_t)(g0 % A.rows) * WV_PH + lane];
    }
    for (int g = blockIdx.x; g < A.total;) {
        // rows are handed out by a device-scope ticket counter (zeroed by the prepare launch), like the workgroup kernels
        int ticket = 0;
        if (lane == 0) ticket = A.work ? (int)gridDim.x + (int)atomicAdd(A.work, 1u) : g + (int)gridDim.x;
        const int gn = __builtin_amdgcn_readfirstlane(ticket);
        const int gc = gn < A.total ? gn : A.total - 1;
        const int b = g / A.rows, r_ = g - b * A.rows;
        const C *__restrict__ ph = phasor + (size_t)r_ * WV_PH;
        const __amdgpu_buffer_rsrc_t rs_spec = __builtin_amdgcn_make_buffer_rsrc((void *)(A.spec + (size_t)b * (2 * 4096)), 0,
                                                                                 2 * 4096 * (int)sizeof(C), 0x00020000);
        const C f0 = conj(lo_n1);              // conj(e^{i ph n1})
        const C f1 = cmul(f0, W.th);           //  ... * e^{+2 pi i n1 / 8192}: the odd chain's half-bin rotation
        C e[64], o[64];
        wave_chain<T, 0>(e, a, ph + 64, f0, W, Lx, rs_spec);
        wave_chain<T, 1>(o, a, ph + 128, f1, W, Lx, rs_spec);

        // ---- last radix-2 stage + |.|^2 + argmax + write-through stores; the next row's needle under it ----
        const __amdgpu_buffer_rsrc_t rs_next = __builtin_amdgcn_make_buffer_rsrc((void *)(A.sig + (size_t)(gc / A.rows) * F_N), 0,
                                                                                 F_N * (int)sizeof(C), 0x00020000);
        T *const out = A.surface ? A.surface + (size_t)g * F_L : nullptr;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(out, 0, out ? F_L * (int)sizeof(T) : 0, 0x00020000);
        T bv_lo = T(0), bv_hi = T(0);
        int bi_lo = 0, bi_hi = 0;
        T mlo[64], mhi[64];
        const C *__restrict__ w128 = A.tab.tw4096;
#pragma unroll
        for (int r = 0; r < 64; ++r) {   // m = lane + 64 r:  E[m] +- W_128^r (W_8192^lane O[m])
            const C wo = r ? cmul_s(o[r], w128[32 * r]) : o[r];   // W_128^r = W_4096^(32 r): a scalar load
            const C lo = e[r] + wo, hi = e[r] - wo;
            mlo[r] = norm_sqr(lo);   // mod.rs:147
            mhi[r] = norm_sqr(hi);
            bi_lo = mlo[r] > bv_lo ? r : bi_lo;   // first strictly greater (mod.rs:148-151)
            bv_lo = vmax(bv_lo, mlo[r]);
            bi_hi = mhi[r] > bv_hi ? r : bi_hi;
            bv_hi = vmax(bv_hi, mhi[r]);
            a[r] = bload(rs_next, (unsigned)(lane * sizeof(C)), (unsigned)(64 * r * sizeof(C)), (C *)nullptr);
        }
        lo_n1 = phasor[(size_t)(gc % A.rows) * WV_PH + lane];
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            typename pair_vec<T>::type dlo, dhi;
            pair_xor1(mlo[2 * j], mlo[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dlo);
            pair_xor1(mhi[2 * j], mhi[2 * j + 1], EVEN_LANES, ~EVEN_LANES, dhi);
            const int m = mpair + 64 * (2 * j + (odd ? 1 : 0));
            store_vec_aux<AUX>(rs, (unsigned)(m * sizeof(T)), dlo);
            store_vec_aux<AUX>(rs, (unsigned)((m + F_N) * sizeof(T)), dhi);
        }
        T bv = bv_lo;
        uint32_t bi = bv_lo > T(0) ? (uint32_t)(lane + 64 * bi_lo) : 0u;
        if (bv_hi > bv) { bv = bv_hi; bi = (uint32_t)(lane + 64 * bi_hi + F_N); }
        wave_arg_reduce_maxmin(bv, bi);
        if (lane == 63) {
            A.row_idx[g] = bi;
            A.row_val[g] = bv;
        }
        g = gn;
    }
}

}  // namespace caf
