// fused1x1.h -- persistent streaming kernel for the short-K 1x1 GEMMs that carry the fused output stage (included by conv.hip).
//
//   O[p][c] = gate_{out_mask}( relu?( acc[p][c] * out_scale[c] + out_shift[c] + bias[c] + res_scale[c] * Res[p][c] ) ),  bits_out = (O > 0)
//   acc = X[P][K] W[Cm][K]^T,  K <= 512,  Cm % 128 == 0
//
// These are conv3 + bn3 + identity + ReLU of every bottleneck (torchvision Bottleneck tail under Encoders.py:336-339), the masked conv1 data
// gradients of its backward, and in the inference forward the same conv3 with the running-statistics coefficients.  Per 128 x 128 output
// tile they move 64 KB of HBM traffic (residual in, result out) for at most 16.8 MFLOP: the launch is a STREAM with a small GEMM riding on
// it, and what the tile-per-workgroup kernels lose is overlap -- every workgroup of a launch starts in the same phase, so the chip
// alternates between "all CUs fetch operands" (HBM idle) and "all CUs run their output stage" (L2 -> LDS path idle); measured 1.3x (layer1)
// to 2.5x (layer4) of the byte roof (profiles/r04_gemm_launches_in_step.md).
//
// Structure: ONE workgroup per CU for the whole launch: 8 consumer waves (MFMA + output stage + stores), 4 ring producers (operand k-steps)
// and 2 residual producers (residual tile + mask bytes).
//   * every global READ is an LDS-DMA (`buffer_load ... lds`) issued by a wave that does nothing else: the operand k-steps go into a 4-stage
//     ring that runs ahead across tile boundaries (while the consumers are in tile t's output stage the first k-steps of tile t + 1 land),
//     the residual tile and the mask bytes of tile t + 1 are requested at the start of tile t's main loop into the other half of a double
//     buffer.  No wave ever waits for an HBM round trip it has just requested.
//   * `vmcnt` retires in issue order per wave: a wave that stored would make its later loads wait for the stores' acknowledgements, and a
//     wave that fetched residuals (an HBM round trip) would hold up the ring's L2 hits behind them.  Hence three roles with a counter each:
//     consumers only store, ring producers and residual producers wait with counted `vmcnt(N)` for exactly the pieces the next beat needs.
//     (Consumers that issued the residual DMA themselves got a compiler-inserted `vmcnt(0)` in front of their first LDS access of the
//     output stage: hipcc orders every LDS access it can see behind the LDS-DMA writes of the same wave.)
//   * the output stage works in place: the residual sits in LDS in the output layout ([pixel][128 channels], 16-byte chunk index XORed
//     with pixel & 15: conflict-free for the accumulator layout's 8-byte accesses and for the 16-byte read-out), the result overwrites it,
//     a barrier, and 16 adjacent lanes store one pixel's 256 contiguous bytes.
//   * one `s_barrier` per beat for all 14 waves (gfx950 has no named barriers): K / 32 main beats + 2 output-stage beats per tile.
//   * tile order: an XCD owns a contiguous range of pixel tiles; its workgroups walk (pixel tile, channel tile) with the channel tile
//     fastest, so the pixel panel is fetched from HBM once per XCD and -- tiles_m dividing the workgroups per XCD -- every workgroup keeps
//     ONE channel tile for the whole launch (its weight rows stay in L2 / the vector cache, its coefficients in LDS).
#pragma once

namespace dali {

// bits 0 / 1: the low / high bf16 half of w is > 0 (as a signed 16-bit integer: sign clear and magnitude non-zero), five integer operations
__device__ __forceinline__ unsigned f1_pos_bits(uint32_t w) {
    const uint32_t p = ((w & 0x7fff7fffu) + 0x7fff7fffu) & ~w & 0x80008000u;      // bit 15 / 31 set iff that half is positive (no carry between halves)
    return ((p >> 15) | (p >> 30)) & 3u;
}

template <int N> __device__ __forceinline__ void f1_vm_wait_imm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
// wave-uniform n; anything beyond the table waits for everything (stricter = safe)
__device__ __forceinline__ void f1_vm_wait(int n) {
    switch (n) {
        case 1: f1_vm_wait_imm<1>(); break;
        case 2: f1_vm_wait_imm<2>(); break;
        case 3: f1_vm_wait_imm<3>(); break;
        case 4: f1_vm_wait_imm<4>(); break;
        case 5: f1_vm_wait_imm<5>(); break;
        case 6: f1_vm_wait_imm<6>(); break;
        case 7: f1_vm_wait_imm<7>(); break;
        case 8: f1_vm_wait_imm<8>(); break;
        case 9: f1_vm_wait_imm<9>(); break;
        case 10: f1_vm_wait_imm<10>(); break;
        case 11: f1_vm_wait_imm<11>(); break;
        case 12: f1_vm_wait_imm<12>(); break;
        case 13: f1_vm_wait_imm<13>(); break;
        case 14: f1_vm_wait_imm<14>(); break;
        case 15: f1_vm_wait_imm<15>(); break;
        case 16: f1_vm_wait_imm<16>(); break;
        case 17: f1_vm_wait_imm<17>(); break;
        case 18: f1_vm_wait_imm<18>(); break;
        case 19: f1_vm_wait_imm<19>(); break;
        case 20: f1_vm_wait_imm<20>(); break;
        case 21: f1_vm_wait_imm<21>(); break;
        case 22: f1_vm_wait_imm<22>(); break;
        case 23: f1_vm_wait_imm<23>(); break;
        case 24: f1_vm_wait_imm<24>(); break;
        case 25: f1_vm_wait_imm<25>(); break;
        case 26: f1_vm_wait_imm<26>(); break;
        case 27: f1_vm_wait_imm<27>(); break;
        case 28: f1_vm_wait_imm<28>(); break;
        default: f1_vm_wait_imm<0>(); break;
    }
}

// the residual producers' counts: 0, 1 (mask only), 16 (residual only), 17 (both)
__device__ __forceinline__ void f1_vm_wait_pieces(int n) {
    if (n == 17) f1_vm_wait_imm<17>();
    else if (n == 16) f1_vm_wait_imm<16>();
    else if (n == 1) f1_vm_wait_imm<1>();
    else f1_vm_wait_imm<0>();
}

constexpr int F1_TM = 128, F1_TN = 128, F1_NC = 8, F1_NP = 4, F1_NR = 2;
constexpr int F1_STAGE_ELEMS = (F1_TM + F1_TN) * 64;                  // one k-step of 64 (full 128-byte lines: the L2 -> LDS path retires lines, not bytes): 32 KB
constexpr int F1_R_BYTES = F1_TN * F1_TM * 2;                         // residual / result tile: 32 KB
constexpr int F1_M_BYTES = F1_TN * F1_TM / 8;                         // mask bytes of a tile: 2 KB
constexpr int F1_C_BYTES = 3 * F1_TM * 4;                             // scale, shift (+ bias), residual scale of the workgroup's channel tile
// NS ring stages, RB residual buffers: <2, 2> for K <= 128 (the main loop is too short to hide a residual fetch: it is requested a tile ahead),
// <3, 1> for K >= 256
// ARES: the workgroup's 128 weight rows (all of K) stay in LDS for the whole launch -- every workgroup keeps ONE channel tile -- and the ring
// carries the pixel rows only: half the DMA pieces per k-step (the ring producers' issue rate, not the MFMA, set the main beats' length)
constexpr int f1_lds_bytes(int NS, int RB, bool ARES, int K) {
    return (ARES ? F1_TM * K * 2 + NS * F1_TN * 64 * 2 : NS * F1_STAGE_ELEMS * 2) + RB * F1_R_BYTES + RB * F1_M_BYTES + F1_C_BYTES;
}

// HAS_RES / HAS_OM / HAS_BITS: residual, output mask, mask bits of the result -- compile-time, so that the two output-stage beats are straight-line
// code (with run-time flags every LDS read sat behind a branch and waited for its own latency: 1.4 + 0.7 us per tile)
// SRC2: the K dimension is the channels of X followed by those of a second plain [P][channels] tensor X2 (IGemmArgs::X2 / Ck1, both multiples of
// 64): conv3 and the downsample convolution of a bottleneck as ONE GEMM in the inference forward ([a2 | x] against [s3.W3 | sd.Wd])
template <int F1_NS, int F1_RB, bool ARES, bool HAS_RES, bool HAS_OM, bool HAS_BITS, bool SRC2 = false>
__global__ __launch_bounds__((F1_NC + F1_NP + F1_NR) * 64) void fused1x1_persist_kernel(IGemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) uint16_t smem[];
    const int K = a.g.Ck, KT = K >> 6, P = a.P, Cm = a.Cm;
    constexpr int STAGE = ARES ? F1_TN * 64 : F1_STAGE_ELEMS;             // elements of a ring stage
    uint16_t* const ring = smem + (ARES ? F1_TM * K : 0);                 // (ARES: the resident weight image [K / 64][128 rows][64] sits in front)
    char* const rbuf = reinterpret_cast<char*>(ring + F1_NS * STAGE);
    char* const mbuf = rbuf + F1_RB * F1_R_BYTES;
    float* const cbuf = reinterpret_cast<float*>(mbuf + F1_RB * F1_M_BYTES);         // [3][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- this workgroup's tiles ----
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, S = gridDim.x >> 3;
    const int n0 = (int)((long long)xcd * tiles_n / 8), n1 = (int)((long long)(xcd + 1) * tiles_n / 8);
    const int items = (n1 - n0) * tiles_m;
    const int T = slot < items ? (items - slot + S - 1) / S : 0;
    if (T == 0) return;
    const int GT = T * KT;                                            // k-steps of the whole launch, numbered through
    auto tile_of = [&](int t, int& tm, int& tn) {
        const int q = slot + S * t, d = q / tiles_m;
        tn = n0 + d;
        tm = q - d * tiles_m;
    };

    if (wave >= F1_NC && wave < F1_NC + F1_NP) {
        // ---------------- ring producers: the operand k-steps, nothing else ----------------
        const int pw = wave - F1_NC;
        const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.W), 0, Cm * K * 2, 0x00020000);
        const int rep = SRC2 && a.x_rep > 1 ? a.x_rep : 1;             // (SRC2) each tensor's channels appear rep times: [X x rep | X2 x rep]
        const int Ck1 = SRC2 ? a.Ck1 : K, Ck2 = K / rep - Ck1;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(a.X), 0, (int)((long long)P * Ck1 * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_x2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(SRC2 ? a.X2 : a.X), 0, (int)((long long)P * (SRC2 ? Ck2 : Ck1) * 2), 0x00020000);
        // a DMA piece = 8 rows x 128 bytes; LDS slot `lane` of piece q = row 8q + (lane >> 3), physical chunk lane & 7 = logical chunk ^ ((row >> 1) & 7)
        // (the k-tile-64 image of igemm_conv_k64_kernel); q = pw + F1_NP i has the parity of pw, so the logical chunk is the same for all of a lane's pieces
        const int r_in = lane >> 3;
        const int kc = (lane & 7) ^ (((pw & 1) << 2) | (r_in >> 1));
        constexpr int NPC = F1_TM / 8 / F1_NP;                        // pieces per producer, operand and k-step: 4
        constexpr int PK = ARES ? NPC : 2 * NPC;                      // DMA pieces per producer and k-step: 4 (pixels only) or 8
        int issued = 0, it = 0, ik = 0, st_issue = 0;
        uint32_t a_off[NPC], b_off[NPC];
        auto set_tile = [&](int t) {
            int tm, tn;
            tile_of(t, tm, tn);
#pragma unroll
            for (int i = 0; i < NPC; ++i) {
                const int m = tm * F1_TM + 8 * (pw + F1_NP * i) + r_in;
                a_off[i] = (uint32_t)(m * K + kc * 8) * 2u;
                const int p = tn * F1_TN + 8 * (pw + F1_NP * i) + r_in;
                if constexpr (SRC2) b_off[i] = p < P ? (uint32_t)p : DMA_OOB;       // the pixel itself: the row pitch differs between the two tensors
                else b_off[i] = p < P ? (uint32_t)(p * K + kc * 8) * 2u : DMA_OOB;
            }
        };
        set_tile(0);
        if constexpr (ARES) {                                         // the weight rows of this workgroup's channel tile, once (older than every ring piece:
            for (int k = 0; k < KT; ++k)                              //  the first counted wait below covers them)
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(smem + k * (F1_TM * 64) + (pw + F1_NP * i) * 512), 16, a_off[i] + (uint32_t)k * 128u, 0, 0, 0);
        }
        auto issue_one = [&]() {
            uint16_t* sa = ring + st_issue * STAGE;
            uint16_t* sb = ARES ? sa : sa + F1_TM * 64;
            st_issue = st_issue == F1_NS - 1 ? 0 : st_issue + 1;
            const uint32_t kb = (uint32_t)ik * 128u;
            if constexpr (!ARES) {
#pragma unroll
                for (int i = 0; i < NPC; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void_ptr)(sa + (pw + F1_NP * i) * 512), 16, a_off[i] + kb, 0, 0, 0);
            }
            if constexpr (SRC2) {                                     // this k-step lies in X (< Ck1) or in X2: wave-uniform
                const int kc0 = ik * 64;
                const bool second = kc0 >= rep * Ck1;
                const int pitch = second ? Ck2 : Ck1, cbase = (second ? (kc0 - rep * Ck1) % Ck2 : kc0 % Ck1) + kc * 8;
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    const uint32_t off = b_off[i] == DMA_OOB ? DMA_OOB : (uint32_t)((int)b_off[i] * pitch + cbase) * 2u;
                    if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x2, (lds_void_ptr)(sb + (pw + F1_NP * i) * 512), 16, off, 0, 0, 0);
                    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (pw + F1_NP * i) * 512), 16, off, 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int i = 0; i < NPC; ++i) {
                    const uint32_t off = b_off[i] == DMA_OOB ? DMA_OOB : b_off[i] + kb;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void_ptr)(sb + (pw + F1_NP * i) * 512), 16, off, 0, 0, 0);
                }
            }
            ++issued;
            if (++ik == KT) {
                ik = 0;
                if (++it < T) set_tile(it);
            }
        };
        auto fill = [&](int next_step) {                             // k-steps below next_step are consumed: stages up to next_step + NS - 1 may be in flight
            const int lim = GT < next_step + F1_NS ? GT : next_step + F1_NS;
            while (issued < lim) issue_one();
        };
        fill(0);
        f1_vm_wait(PK * (issued - 1));                                // k-step 0 has landed
        __builtin_amdgcn_s_barrier();
        int g = 0;
        const bool stamp = a.stamps != nullptr && pw == 0;             // diagnostic (scripts/fused1x1_timeline.py): ticks spent issuing / waiting / at barriers
        unsigned long long t_issue = 0, t_wait = 0, t_bar = 0;
        for (int t = 0; t < T; ++t) {
            for (int k = 0; k < KT; ++k, ++g) {
                const unsigned long long s0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
                fill(g);
                const unsigned long long s1 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
                if (k + 1 < KT) f1_vm_wait(PK * (issued - 2 - g));    // k-step g + 1 has landed before the consumers reach it
                const unsigned long long s2 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
                __builtin_amdgcn_s_barrier();
                if (stamp) { t_issue += s1 - s0; t_wait += s2 - s1; t_bar += __builtin_amdgcn_s_memrealtime() - s2; }
            }
            fill(g);                                                  // the output stage uses no ring stage: one more k-step fits
            __builtin_amdgcn_s_barrier();
            if (g < GT) f1_vm_wait(PK * (issued - 1 - g));            // the next tile's first k-step has landed
            __builtin_amdgcn_s_barrier();
        }
        if (stamp && lane == 0) {
            a.stamps[(size_t)blockIdx.x * 12 + 9] = t_issue; a.stamps[(size_t)blockIdx.x * 12 + 10] = t_wait; a.stamps[(size_t)blockIdx.x * 12 + 11] = t_bar;
        }
        return;
    }

    constexpr bool has_res = HAS_RES, has_om = HAS_OM, has_bits = HAS_BITS;
    if (wave >= F1_NC + F1_NP) {
        // ---------------- residual producers: the residual tile and the mask bytes of tile t + 1 while tile t is multiplied ----------------
        const int rw = wave - F1_NC - F1_NP;
        const __amdgpu_buffer_rsrc_t rs_r = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint16_t*>(has_res ? a.Res : a.X), 0, has_res ? (int)((long long)P * Cm * 2) : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_m = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(has_om ? a.out_mask : reinterpret_cast<const uint8_t*>(a.X)), 0,
                                                                              has_om ? (int)((long long)P * Cm / 8) : 0, 0x00020000);
        constexpr int RP = F1_R_BYTES / 1024 / F1_NR;                 // residual pieces per wave and tile: 16
        const int pieces = (has_res ? RP : 0) + (has_om ? 1 : 0);
        auto issue_rm = [&](int t) {
            int tm, tn;
            tile_of(t, tm, tn);
            char* rb = rbuf + (F1_RB == 2 ? (t & 1) : 0) * F1_R_BYTES;
            if (has_res) {
#pragma unroll
                for (int i = 0; i < RP; ++i) {
                    const int r = rw + F1_NR * i;                     // piece r: pixels 4r .. 4r + 3, one 256-byte row each
                    const int pix = 4 * r + (lane >> 4);
                    const int lc = (lane & 15) ^ (pix & 15);          // logical chunk behind this lane's physical slot
                    const int p = tn * F1_TN + pix;
                    const uint32_t off = p < P ? (uint32_t)(p * Cm + tm * F1_TM + lc * 8) * 2u : DMA_OOB;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_r, (lds_void_ptr)(rb + r * 1024), 16, off, 0, 0, 0);
                }
            }
            if (has_om) {
                const int p = tn * F1_TN + 64 * rw + lane;            // one pixel's 16 mask bytes per lane
                const uint32_t off = p < P ? (uint32_t)(((long long)p * Cm + tm * F1_TM) >> 3) : DMA_OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_m, (lds_void_ptr)(mbuf + (F1_RB == 2 ? (t & 1) : 0) * F1_M_BYTES + rw * 1024), 16, off, 0, 0, 0);
            }
        };
        issue_rm(0);
        __builtin_amdgcn_s_barrier();
        for (int t = 0; t < T; ++t) {
            for (int k = 0; k < KT; ++k) {
                if constexpr (F1_RB == 2) {
                    if (k == 0 && t + 1 < T) issue_rm(t + 1);
                    // last main beat: tile t's pieces (requested one tile ago) have landed; only tile t + 1's may still be in flight
                    if (k == KT - 1) { if (t + 1 < T) f1_vm_wait_pieces(pieces); else f1_vm_wait_imm<0>(); }
                } else {                                              // one buffer: free since the previous tile's output stage; lands under this main loop
                    if (k == 0 && t > 0) issue_rm(t);
                    if (k == KT - 1) f1_vm_wait_imm<0>();
                }
                __builtin_amdgcn_s_barrier();
            }
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_s_barrier();
        }
        return;
    }

    // ---------------- consumers ----------------
    const int wm = wave >> 2, wn = wave & 3;                          // 2 x 4 waves of 64 channels x 32 pixels
    // the coefficients of this workgroup's channel tile (the launcher guarantees S % tiles_m == 0: the tile never changes), in LDS: held in
    // registers they were 48 VGPRs and the kernel spilled
    {
        int tm, tn;
        tile_of(0, tm, tn);
        if (tid < F1_TM) {
            const int c = tm * F1_TM + tid;
            cbuf[tid] = a.out_scale ? a.out_scale[c] : 1.f;
            cbuf[F1_TM + tid] = (a.out_shift ? a.out_shift[c] : 0.f) + (a.bias ? a.bias[c] : 0.f);
            cbuf[2 * F1_TM + tid] = a.res_scale ? a.res_scale[c] : 1.f;
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // (before any DMA is counted; the first barrier publishes the writes)
    }
    const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ ((lane & 15) >> 1)) << 3);      // first 32 k; the second 32 are physical chunk ^ 4
    const int relu = a.out_relu;
    __builtin_amdgcn_s_barrier();
    int st_cur = 0;
    const bool stamp = a.stamps != nullptr && tid == 0;               // diagnostic: ticks (100 MHz) per phase, summed over this workgroup's tiles
    unsigned long long c_main = 0, c_mainbar = 0, c_e0 = 0, c_e0bar = 0, c_e1 = 0, c_e1bar = 0;
    const unsigned long long c_start = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
    for (int t = 0; t < T; ++t) {
        int tm, tn;
        tile_of(t, tm, tn);
        f32x4_t acc[4][2];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < KT; ++k) {
            const unsigned long long s0 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
            const uint16_t* sb = ring + st_cur * STAGE + (ARES ? 0 : F1_TM * 64);
            const uint16_t* sa = ARES ? smem + k * (F1_TM * 64) : ring + st_cur * STAGE;
            st_cur = st_cur == F1_NS - 1 ? 0 : st_cur + 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int fo = frag_off ^ (h << 5);
                bf16x8_t fa[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[i] = *reinterpret_cast<const bf16x8_t*>(sa + (wm * 64 + i * 16) * 64 + fo);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const bf16x8_t fb = *reinterpret_cast<const bf16x8_t*>(sb + (wn * 32 + j * 16) * 64 + fo);
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb, acc[i][j], 0, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const unsigned long long s1 = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
            __builtin_amdgcn_s_barrier();
            if (stamp) { c_main += s1 - s0; c_mainbar += __builtin_amdgcn_s_memrealtime() - s1; }
        }
        const unsigned long long e0s = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
        // ---- output stage, beat 1: value = acc * scale + shift (+ res_scale * residual) -> ReLU -> bf16, in place over the residual ----
        char* rb = rbuf + (F1_RB == 2 ? (t & 1) : 0) * F1_R_BYTES;
        {
            // all LDS reads first (residual slots, coefficients), then the arithmetic, then all writes: the in-place writes may alias the reads as
            // far as the compiler can tell, so interleaved they serialised the tile's eight LDS round trips (1.2 us per tile)
            // (slot addresses are recomputed, not kept in an array of pointers: that array went to scratch memory in some instantiations)
            auto slot_of = [&](int i, int j) -> uint2* {
                const int pix = wn * 32 + j * 16 + (lane & 15);
                const int chunk = wm * 8 + i * 2 + (lane >> 5);
                return reinterpret_cast<uint2*>(rb + pix * 256 + ((lane >> 4) & 1) * 8 + ((chunk ^ (pix & 15)) << 4));
            };
            const float lo = relu ? 0.f : -__builtin_huge_valf();         // ReLU without a branch: max(v, lo)
#pragma unroll
            for (int j = 0; j < 2; ++j) {                                 // one 16-pixel column block at a time: its four residual slots are read
                uint2 rv[4];                                              // together (one LDS round trip), then overwritten in place
                if (has_res) {
#pragma unroll
                    for (int i = 0; i < 4; ++i) rv[i] = *slot_of(i, j);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int c = wm * 64 + i * 16 + (lane >> 4) * 4;
                    const float4 s4 = *reinterpret_cast<const float4*>(cbuf + c), h4 = *reinterpret_cast<const float4*>(cbuf + F1_TM + c);
                    float4 r4 = make_float4(1.f, 1.f, 1.f, 1.f);
                    if (has_res) r4 = *reinterpret_cast<const float4*>(cbuf + 2 * F1_TM + c);
                    // explicit fused multiply-adds (the library is built with -ffp-contract=off): the output stage is VALU-issue bound, ~5 -> ~3.5
                    // instructions per element; one rounding fewer than the tile-per-workgroup kernels' mul + add, same value on exact data
                    float v0 = __builtin_fmaf(acc[i][j][0], s4.x, h4.x), v1 = __builtin_fmaf(acc[i][j][1], s4.y, h4.y);
                    float v2 = __builtin_fmaf(acc[i][j][2], s4.z, h4.z), v3 = __builtin_fmaf(acc[i][j][3], s4.w, h4.w);
                    if (has_res) {
                        v0 = __builtin_fmaf(r4.x, bf16_bits_to_f32(rv[i].x & 0xffffu), v0); v1 = __builtin_fmaf(r4.y, bf16_bits_to_f32(rv[i].x >> 16), v1);
                        v2 = __builtin_fmaf(r4.z, bf16_bits_to_f32(rv[i].y & 0xffffu), v2); v3 = __builtin_fmaf(r4.w, bf16_bits_to_f32(rv[i].y >> 16), v3);
                    }
                    *slot_of(i, j) = make_uint2(pack_bf16x2(fmaxf(v0, lo), fmaxf(v1, lo)), pack_bf16x2(fmaxf(v2, lo), fmaxf(v3, lo)));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const unsigned long long e0e = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
        __builtin_amdgcn_s_barrier();
        const unsigned long long e1s = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
        // ---- beat 2: 16 adjacent lanes store one pixel's 256 contiguous bytes (+ 16 mask-bit bytes) ----
        const bool interior = (tn + 1) * F1_TN <= P;
        const char* mb = mbuf + (F1_RB == 2 ? (t & 1) : 0) * F1_M_BYTES;
        if (interior) {
            // all four LDS reads first (one round trip), then gate / bits / stores.  (Four named values: as an array indexed in two loops the
            // compiler kept a copy of it in scratch memory -- dead stores, but vector-memory operations of the consumers.)
            auto ld = [&](int itr) { const int s = itr * (F1_NC * 64) + tid, pix = s >> 4, chunk = s & 15;
                                     return *reinterpret_cast<const uint4*>(rb + pix * 256 + ((chunk ^ (pix & 15)) << 4)); };
            auto ldm = [&](int itr) -> unsigned { const int s = itr * (F1_NC * 64) + tid, pix = s >> 4, chunk = s & 15;
                                                   return has_om ? *reinterpret_cast<const uint8_t*>(mb + pix * 16 + chunk) : 0u; };
            auto put = [&](int itr, uint4 v, unsigned m) {
                const int s = itr * (F1_NC * 64) + tid, pix = s >> 4, chunk = s & 15;
                if (has_om) { v.x = gate_bf16x2(v.x, m); v.y = gate_bf16x2(v.y, m >> 2); v.z = gate_bf16x2(v.z, m >> 4); v.w = gate_bf16x2(v.w, m >> 6); }
                const size_t e = ((size_t)(tn * F1_TN + pix) * Cm + tm * F1_TM + chunk * 8);
                *reinterpret_cast<uint4*>(a.O + e) = v;
                if (has_bits) a.bits_out[e >> 3] = (uint8_t)(f1_pos_bits(v.x) | (f1_pos_bits(v.y) << 2) | (f1_pos_bits(v.z) << 4) | (f1_pos_bits(v.w) << 6));
            };
            if constexpr (has_res && has_om) {                         // (the masked-gradient instantiation sits at the 128-register cap: two round trips of two)
                { const uint4 v0 = ld(0), v1 = ld(1); const unsigned k0 = ldm(0), k1 = ldm(1); put(0, v0, k0); put(1, v1, k1); }
                { const uint4 v2 = ld(2), v3 = ld(3); const unsigned k2 = ldm(2), k3 = ldm(3); put(2, v2, k2); put(3, v3, k3); }
            } else {
            const uint4 v0 = ld(0), v1 = ld(1), v2 = ld(2), v3 = ld(3);
            const unsigned k0 = ldm(0), k1 = ldm(1), k2 = ldm(2), k3 = ldm(3);
            put(0, v0, k0); put(1, v1, k1); put(2, v2, k2); put(3, v3, k3);
            }
        } else {
#pragma unroll
            for (int itr = 0; itr < 4; ++itr) {
                const int s = itr * (F1_NC * 64) + tid, pix = s >> 4, chunk = s & 15;
                uint4 v = *reinterpret_cast<const uint4*>(rb + pix * 256 + ((chunk ^ (pix & 15)) << 4));
                if (has_om) {
                    const unsigned m = *reinterpret_cast<const uint8_t*>(mb + pix * 16 + chunk);
                    v.x = gate_bf16x2(v.x, m); v.y = gate_bf16x2(v.y, m >> 2); v.z = gate_bf16x2(v.z, m >> 4); v.w = gate_bf16x2(v.w, m >> 6);
                }
                if (tn * F1_TN + pix < P) {
                    const size_t e = ((size_t)(tn * F1_TN + pix) * Cm + tm * F1_TM + chunk * 8);
                    *reinterpret_cast<uint4*>(a.O + e) = v;
                    if (has_bits) a.bits_out[e >> 3] = (uint8_t)(f1_pos_bits(v.x) | (f1_pos_bits(v.y) << 2) | (f1_pos_bits(v.z) << 4) | (f1_pos_bits(v.w) << 6));
                }
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this tile's LDS reads are done before its buffer is refilled (tile t + 2's pieces)
        const unsigned long long e1e = stamp ? __builtin_amdgcn_s_memrealtime() : 0;
        __builtin_amdgcn_s_barrier();
        if (stamp) { c_e0 += e0e - e0s; c_e0bar += e1s - e0e; c_e1 += e1e - e1s; c_e1bar += __builtin_amdgcn_s_memrealtime() - e1e; }
    }
    if (stamp) {
        unsigned long long* o = a.stamps + (size_t)blockIdx.x * 12;
        o[0] = c_start; o[3] = __builtin_amdgcn_s_memrealtime(); o[1] = c_main; o[2] = c_mainbar; o[4] = c_e0; o[5] = c_e0bar; o[6] = c_e1; o[7] = c_e1bar; o[8] = (unsigned long long)T;
    }
}

}  // namespace dali
