// Complex helpers and the small in-register DFTs (radix 2, 3, 4, 5, 6, 8, 9, 10) the FFT passes are built from (pse_kernels.hip: x and y
// passes, the z pass at 256 / 512; pse_zfft.hip: the z pass at the other 2^a 3^b 5^c sizes).  Device code only.
#pragma once
#include <hip/hip_runtime.h>

namespace pse {

__device__ __forceinline__ double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

template <int R, bool INVERSE, int LEN>
__device__ __forceinline__ void dft_small(double2 (&v)[LEN]) {
    static_assert(LEN >= R, "the butterfly's points");
    const double sg = INVERSE ? 1.0 : -1.0;                          // forward: exp(-i ...)
    if constexpr (R == 2) {
        const double2 a = v[0], b = v[1];
        v[0] = make_double2(a.x + b.x, a.y + b.y); v[1] = make_double2(a.x - b.x, a.y - b.y);
    } else if constexpr (R == 3) {
        constexpr double S = 0.86602540378443864676;                 // sin(2 pi / 3)
        const double2 s = make_double2(v[1].x + v[2].x, v[1].y + v[2].y), t = make_double2(v[1].x - v[2].x, v[1].y - v[2].y);
        const double2 m = make_double2(v[0].x - 0.5 * s.x, v[0].y - 0.5 * s.y);
        const double2 jt = make_double2(-sg * S * t.y, sg * S * t.x);   // (+-i) sin (v1 - v2)
        v[0] = make_double2(v[0].x + s.x, v[0].y + s.y);
        v[1] = make_double2(m.x + jt.x, m.y + jt.y);
        v[2] = make_double2(m.x - jt.x, m.y - jt.y);
    } else if constexpr (R == 4) {
        const double2 a = v[0], b = v[1], c = v[2], d = v[3];
        const double2 s0 = make_double2(a.x + c.x, a.y + c.y), s1 = make_double2(a.x - c.x, a.y - c.y);
        const double2 s2 = make_double2(b.x + d.x, b.y + d.y), s3 = make_double2(b.x - d.x, b.y - d.y);
        const double2 j3 = make_double2(-sg * s3.y, sg * s3.x);
        v[0] = make_double2(s0.x + s2.x, s0.y + s2.y); v[1] = make_double2(s1.x + j3.x, s1.y + j3.y);
        v[2] = make_double2(s0.x - s2.x, s0.y - s2.y); v[3] = make_double2(s1.x - j3.x, s1.y - j3.y);
    } else if constexpr (R == 8) {
        // 8 = 2 x 4: X[k1 + 2 k2] = sum_n2 W8^{n2 k1} W4^{n2 k2} (x[n2] + (-1)^k1 x[n2 + 4])
        constexpr double H = 0.70710678118654752440;
        double2 e[4], o[4];
#pragma unroll
        for (int n2 = 0; n2 < 4; ++n2) {
            e[n2] = make_double2(v[n2].x + v[n2 + 4].x, v[n2].y + v[n2 + 4].y);
            o[n2] = make_double2(v[n2].x - v[n2 + 4].x, v[n2].y - v[n2 + 4].y);
        }
        // W8^{n2} on the odd branch: 1, (1 -+ i) / sqrt2, -+i, (-1 -+ i) / sqrt2   (forward: upper signs)
        o[1] = make_double2(H * (o[1].x - sg * o[1].y), H * (o[1].y + sg * o[1].x));
        o[2] = make_double2(-sg * o[2].y, sg * o[2].x);
        o[3] = make_double2(H * (-o[3].x - sg * o[3].y), H * (-o[3].y + sg * o[3].x));
        auto dft4 = [&](double2 (&q)[4]) __attribute__((always_inline)) {
            const double2 s0 = make_double2(q[0].x + q[2].x, q[0].y + q[2].y), s1 = make_double2(q[0].x - q[2].x, q[0].y - q[2].y);
            const double2 s2 = make_double2(q[1].x + q[3].x, q[1].y + q[3].y), s3 = make_double2(q[1].x - q[3].x, q[1].y - q[3].y);
            const double2 j3 = make_double2(-sg * s3.y, sg * s3.x);
            q[0] = make_double2(s0.x + s2.x, s0.y + s2.y); q[1] = make_double2(s1.x + j3.x, s1.y + j3.y);
            q[2] = make_double2(s0.x - s2.x, s0.y - s2.y); q[3] = make_double2(s1.x - j3.x, s1.y - j3.y);
        };
        dft4(e); dft4(o);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) { v[2 * k2] = e[k2]; v[2 * k2 + 1] = o[k2]; }
    } else if constexpr (R == 9) {
        // 9 = 3 x 3: X[k1 + 3 k2] = sum_n2 W9^{n2 k1} W3^{n2 k2} sum_n1 x[n2 + 3 n1] W3^{n1 k1}
        constexpr double S = 0.86602540378443864676;
        constexpr double C91 = 0.76604444311897803520, S91 = 0.64278760968653932632;    // cos, sin (2 pi / 9)
        constexpr double C92 = 0.17364817766693034885, S92 = 0.98480775301220805937;    // (4 pi / 9)
        constexpr double C94 = -0.93969262078590838405, S94 = 0.34202014332566873304;   // (8 pi / 9)
        auto dft3 = [&](double2 &a, double2 &b, double2 &c) __attribute__((always_inline)) {
            const double2 s = make_double2(b.x + c.x, b.y + c.y), t = make_double2(b.x - c.x, b.y - c.y);
            const double2 m = make_double2(a.x - 0.5 * s.x, a.y - 0.5 * s.y);
            const double2 jt = make_double2(-sg * S * t.y, sg * S * t.x);
            a = make_double2(a.x + s.x, a.y + s.y);
            b = make_double2(m.x + jt.x, m.y + jt.y);
            c = make_double2(m.x - jt.x, m.y - jt.y);
        };
        auto rot = [&](double2 &q, double cw, double sw) __attribute__((always_inline)) {   // q *= cos + sg i sin
            q = make_double2(q.x * cw - sg * q.y * sw, q.y * cw + sg * q.x * sw);
        };
        // inner transforms over n1 for n2 = 0, 1, 2: t[n2][k1] lands in v[n2 + 3 k1]
        dft3(v[0], v[3], v[6]); dft3(v[1], v[4], v[7]); dft3(v[2], v[5], v[8]);
        rot(v[4], C91, S91); rot(v[7], C92, S92);        // n2 = 1: W9^{k1}
        rot(v[5], C92, S92); rot(v[8], C94, S94);        // n2 = 2: W9^{2 k1}
        // outer transforms over n2 for k1 = 0, 1, 2: X[k1 + 3 k2] lands in (the slot of n2 = k2) v[k2 + 3 k1]
        dft3(v[0], v[1], v[2]); dft3(v[3], v[4], v[5]); dft3(v[6], v[7], v[8]);
        // natural order: X[k1 + 3 k2] = v[k2 + 3 k1]
        const double2 x1 = v[3], x2 = v[6], x3 = v[1], x5 = v[7], x6 = v[2], x7 = v[5];
        v[1] = x1; v[2] = x2; v[3] = x3; v[5] = x5; v[6] = x6; v[7] = x7;
    } else if constexpr (R == 6 || R == 10) {
        // 2 x Q with Q = 3 or 5 (coprime: Good-Thomas, no twiddles): x[(Q n1 + 2 n2) mod R] -> X[(Q k1 + (Q + 1) k2) mod R]
        constexpr int Q = R / 2;
        double2 sm[Q], df[Q];
#pragma unroll
        for (int n2 = 0; n2 < Q; ++n2) {
            const double2 a = v[(2 * n2) % R], b = v[(Q + 2 * n2) % R];
            sm[n2] = make_double2(a.x + b.x, a.y + b.y); df[n2] = make_double2(a.x - b.x, a.y - b.y);
        }
        dft_small<Q, INVERSE>(sm); dft_small<Q, INVERSE>(df);
#pragma unroll
        for (int k2 = 0; k2 < Q; ++k2) { v[((Q + 1) * k2) % R] = sm[k2]; v[(Q + (Q + 1) * k2) % R] = df[k2]; }
    } else {
        static_assert(R == 5, "radix");
        constexpr double C1 = 0.30901699437494742410, C2 = -0.80901699437494742410;   // cos(2 pi/5), cos(4 pi/5)
        constexpr double S1 = 0.95105651629515357212, S2 = 0.58778525229247312917;    // sin(2 pi/5), sin(4 pi/5)
        const double2 a1 = make_double2(v[1].x + v[4].x, v[1].y + v[4].y), b1 = make_double2(v[1].x - v[4].x, v[1].y - v[4].y);
        const double2 a2 = make_double2(v[2].x + v[3].x, v[2].y + v[3].y), b2 = make_double2(v[2].x - v[3].x, v[2].y - v[3].y);
        const double2 m1 = make_double2(v[0].x + C1 * a1.x + C2 * a2.x, v[0].y + C1 * a1.y + C2 * a2.y);
        const double2 m2 = make_double2(v[0].x + C2 * a1.x + C1 * a2.x, v[0].y + C2 * a1.y + C1 * a2.y);
        const double2 t1 = make_double2(S1 * b1.x + S2 * b2.x, S1 * b1.y + S2 * b2.y);
        const double2 t2 = make_double2(S2 * b1.x - S1 * b2.x, S2 * b1.y - S1 * b2.y);
        const double2 j1 = make_double2(-sg * t1.y, sg * t1.x), j2 = make_double2(-sg * t2.y, sg * t2.x);
        v[0] = make_double2(v[0].x + a1.x + a2.x, v[0].y + a1.y + a2.y);
        v[1] = make_double2(m1.x + j1.x, m1.y + j1.y); v[4] = make_double2(m1.x - j1.x, m1.y - j1.y);
        v[2] = make_double2(m2.x + j2.x, m2.y + j2.y); v[3] = make_double2(m2.x - j2.x, m2.y - j2.y);
    }
}

}  // namespace pse
