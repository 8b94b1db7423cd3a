p='/tmp/diag/kernels.hpp'  # a scratch copy of camera-calibration_amd/csrc/kernels.hpp
s=open(p).read()
def rep(old,new,count=1):
    global s
    assert old in s, old[:80]
    s=s.replace(old,new,count)
rep('''constexpr int kFusedRowChunks = 17;''','''#ifdef CALIB_STAMPS
// diagnostic build only: per-wave s_memtime deltas of the fused kernel's phases (tools/diag/)
constexpr int kStampWaves = 1 << 19;
__device__ unsigned long long g_stamps[kStampWaves * 10];
#define STAMP(i) do { __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_sched_barrier(0); const unsigned long long t__ = __builtin_amdgcn_s_memtime(); tacc[i] += t__ - tlast; tlast = t__; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define STAMP(i) do {} while (0)
#endif
constexpr int kFusedRowChunks = 17;''')
rep('''    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES * SLAB * sizeof(T2)];
    if (sel && st->done) return;''','''    __shared__ __attribute__((aligned(16))) unsigned char smem[WAVES * SLAB * sizeof(T2)];
#ifdef CALIB_STAMPS
    unsigned long long tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_amdgcn_s_memtime();
#endif
#ifndef CALIB_IGNORE_DONE
    if (sel && st->done) return;
#endif''')
rep('''        for (;; ++item) {
        const T* vc =''','''        STAMP(0);
        for (;; ++item) {
        const T* vc =''')
rep('''            if (q0 + 64 < qend) {
                pn = pbeg + (q + 64 < qend ? q + 64 : qend - 1);
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
            } else if (MULTI && item < item_last) {           // the next item's first batch (uniform items: qbeg = 0)
                pn = pbeg + n + (sl < n ? sl : n - 1);
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
            }''','''            if (q0 + 64 < qend) {
                pn = pbeg + (q + 64 < qend ? q + 64 : qend - 1);
#ifndef CALIB_ABLATE_GLD
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
#endif
#ifdef CALIB_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F73);     // vmcnt(3): this batch's points have arrived
#endif
            } else if (MULTI && item < item_last) {           // the next item's first batch (uniform items: qbeg = 0)
                pn = pbeg + n + (sl < n ? sl : n - 1);
#ifndef CALIB_ABLATE_GLD
                m_n = uv[pn]; xy_n = XY[pn]; z_n = Z[pn];
#endif
#ifdef CALIB_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F73);
#endif
            } else {
#ifdef CALIB_STAMPS
                __builtin_amdgcn_s_waitcnt(0x0F70);     // vmcnt(0)
#endif
            }
            STAMP(1);''')
rep('''        int64_t pn = pbeg + (qbeg + sl < qend ? qbeg + sl : qend - 1);
        T2 m_n = uv[pn], xy_n = XY[pn];
        T z_n = Z[pn];''','''        int64_t pn = pbeg + (qbeg + sl < qend ? qbeg + sl : qend - 1);
#ifdef CALIB_ABLATE_GLD
        T2 m_n, xy_n; m_n.x = T(lane); m_n.y = T(1); xy_n.x = T(0.01) * T(lane); xy_n.y = T(0.02);
        T z_n = T(0);
#else
        T2 m_n = uv[pn], xy_n = XY[pn];
        T z_n = Z[pn];
#endif''')
rep('''            jacobian_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v, Jc);
            // lanes past the item's end evaluate a clamped (finite) point; their rows are only ever''','''#ifdef CALIB_ABLATE_VALU
            u = xy.x; v = xy.y;
#pragma unroll
            for (int cc = 0; cc < C; ++cc) { Jc[cc].x = xy.x; Jc[cc].y = xy.y; }
            asm volatile("" :: "v"(z));
#else
            jacobian_point<MODEL, T>(sp, vc, xy.x, xy.y, z, u, v, Jc);
#endif
            // lanes past the item's end evaluate a clamped (finite) point; their rows are only ever''')
rep('''            T2 res;
            res.x = m.x - u;
            res.y = m.y - v;
''','''            T2 res;
            res.x = m.x - u;
            res.y = m.y - v;
            STAMP(2);
''')
rep('''                T2* row = slab + rowOff(sl & (ROWS - 1));
                if (HALVES == 1 || (lane >> 5) == half) {
                    T* rh = reinterpret_cast<T*>(row);
                    rh[0] = Jc[0].x; rh[3] = Jc[1].y; rh[4] = Jc[2].x;  // the non-zero halves of columns 0, 1, 2
                }
                if constexpr (HALVES == 2) {''','''                T2* row = slab + rowOff(sl & (ROWS - 1));
#ifndef CALIB_ABLATE_LDSW
                if (HALVES == 1 || (lane >> 5) == half) {
                    T* rh = reinterpret_cast<T*>(row);
                    rh[0] = Jc[0].x; rh[3] = Jc[1].y; rh[4] = Jc[2].x;  // the non-zero halves of columns 0, 1, 2
                }
#endif
#ifdef CALIB_ABLATE_LDSW
                if constexpr (false) {
#else
                if constexpr (HALVES == 2) {
#endif''')
rep('''                } else {
#pragma unroll
                    for (int i = 0; i < NCH; ++i) row[colOf(i)] = ch[i];
                }''','''                } else {
#ifndef CALIB_ABLATE_LDSW
#pragma unroll
                    for (int i = 0; i < NCH; ++i) row[colOf(i)] = ch[i];
#else
                    asm volatile("" :: "v"(ch[0].x), "v"(ch[NCH - 1].y));
#endif
                }''')
rep('''                __builtin_amdgcn_wave_barrier();
                const int rows = qend - (q0 + ROWS * half);     // valid points in this pass (may exceed ROWS)''','''                __builtin_amdgcn_wave_barrier();
                STAMP(3);
                const int rows = qend - (q0 + ROWS * half);     // valid points in this pass (may exceed ROWS)''')
rep('''                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.x, (double)ja.x, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.y, (double)ja.y, acc2, 0, 0, 0);
                    }
                    if (rows < ROWS && (rows & 3)) {''','''#ifdef CALIB_ABLATE_MFMA
                        asm volatile("" :: "v"(ja.x), "v"(ja.y));
#else
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.x, (double)ja.x, acc, 0, 0, 0);
                        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)ja.y, (double)ja.y, acc2, 0, 0, 0);
#endif
                    }
                    if (rows < ROWS && (rows & 3)) {''')
rep('''                        for (int s = 0; s < ROWS / 4; ++s) {
                            fu = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].x, jv[s].x, fu, 0, 0, 0);
                            fv = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].y, jv[s].y, fv, 0, 0, 0);
                        }''','''                        for (int s = 0; s < ROWS / 4; ++s) {
#ifdef CALIB_ABLATE_MFMA
                            asm volatile("" :: "v"(jv[s].x), "v"(jv[s].y));
#else
                            fu = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].x, jv[s].x, fu, 0, 0, 0);
                            fv = __builtin_amdgcn_mfma_f32_16x16x4f32(jv[s].y, jv[s].y, fv, 0, 0, 0);
#endif
                        }''')
rep('''                    auto contract = [&](const T2& ja, const T2& jb, double ha, double hc) {
                        d0u =''','''                    auto contract = [&](const T2& ja, const T2& jb, double ha, double hc) {
#ifdef CALIB_ABLATE_MFMA
                        asm volatile("" :: "v"(ja.x), "v"(ja.y), "v"(jb.x), "v"(jb.y), "v"(ha), "v"(hc));
                        return;
#endif
                        d0u =''')
rep('''                            contract(ja, jb, ha, hc);
                        }
                    }
                }
            }
        }
        if (!MULTI || item >= item_last) break;''','''                            contract(ja, jb, ha, hc);
                        }
                    }
                }
                STAMP(4);
            }
        }
        if (!MULTI || item >= item_last) break;''')
rep('''    if (lane == 0) { TU[kEmitZero] = 0.0; TV[kEmitZero] = 0.0; }
    __syncthreads();''','''    if (lane == 0) { TU[kEmitZero] = 0.0; TV[kEmitZero] = 0.0; }
    STAMP(5);
    __syncthreads();
    STAMP(6);''')
rep('''    if ((!G44 && wpi == 1) || sub != 0 || !valid) return;       // the item's first wave assembles the record from the parked tiles''','''    STAMP(7);
#ifdef CALIB_STAMPS
    if ((!G44 && wpi == 1) || sub != 0 || !valid) {
        // (the direct record stores of the 16x16x4 / fp32 forms were issued before the tile parking: they are in "park tiles")
        if (lane == 0 && valid) { const int wid = (blockIdx.x * WAVES + wave) & (kStampWaves - 1); tacc[9] = 1; for (int i = 0; i < 10; ++i) g_stamps[(size_t)wid * 10 + i] = tacc[i]; }
        return;
    }
#else
    if ((!G44 && wpi == 1) || sub != 0 || !valid) return;       // the item's first wave assembles the record from the parked tiles
#endif''')
rep('''        *reinterpret_cast<double2*>(G + i0) = o;
    }
}''','''        *reinterpret_cast<double2*>(G + i0) = o;
    }
#ifdef CALIB_STAMPS
    __builtin_amdgcn_s_waitcnt(0);
    STAMP(8);
    if (lane == 0) { const int wid = (blockIdx.x * WAVES + wave) & (kStampWaves - 1); tacc[9] = 1; for (int i = 0; i < 10; ++i) g_stamps[(size_t)wid * 10 + i] = tacc[i]; }
#endif
}''')
open(p,'w').write(s)
p='/tmp/diag/calib_lm.hip'
s=open(p).read()
s+='''
#ifdef CALIB_STAMPS
// diagnostic build only: sums over the waves that stamped (slot 9 of a wave = 1), then clears
extern "C" int calib_debug_stamps(double* out10) {
    std::vector<unsigned long long> h((size_t)calib::kStampWaves * 10);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(calib::g_stamps), h.size() * 8);
    for (int i = 0; i < 10; ++i) out10[i] = 0.0;
    for (size_t w = 0; w < (size_t)calib::kStampWaves; ++w)
        if (h[w * 10 + 9]) for (int i = 0; i < 10; ++i) out10[i] += (double)h[w * 10 + i];
    std::fill(h.begin(), h.end(), 0ull);
    (void)hipMemcpyToSymbol(HIP_SYMBOL(calib::g_stamps), h.data(), h.size() * 8);
    return 0;
}
#endif
'''
open(p,'w').write(s)
print("ok")
