// Hand-scheduled inner-node loop of the wave-packet LBVH walk for gfx950.
//
// One call walks inner nodes from `cur` until a leaf reference comes up (returned,
// negative) or the wave-shared stack is empty (PACKET_DONE).  The loop is written in
// assembly because its speed is set by instruction ISSUE, scalar as much as vector:
// per node it spends 18 vector instructions (6 packed fused multiply-adds for the 12
// plane distances of the two child boxes, then max3 / min3 / clamp / widen / bound /
// compare per box) and about 15 scalar ones (address, one 64-byte scalar load, the two
// ballots masked with the live lanes, the majority vote for the nearer child, push or
// pop).  The compiler's version of the same loop spends about 45 scalar instructions
// per node on materialising and re-testing its branch conditions.
//
// Register contract (hard-coded temporaries are listed as clobbers):
//   s[60:75]  the node record          s[76:87]  masks, vote counts, saved m0, offset
//   v[8:19]   the six (t_lo, t_hi) pairs   v[20:29]  near / far reductions
// Operands: A = (inv.x, inv.y), B = (inv.z, -o.x*inv.x), C = (-o.y*inv.y, -o.z*inv.z);
// the packed multiply-adds pick the half they need with op_sel, so the six per-ray
// constants occupy six registers.  The stack holds at most 64 entries (entry k in lane k
// of `vs`): callers use this loop only for trees whose depth is at most 64.
#pragma once

#define PKA_STR2(x) #x
#define PKA_STR(x) PKA_STR2(x)

// near / far registers of the two boxes per axis, for a direction that is positive
// (lo plane first) or negative (hi plane first) along that axis: near0, far0, near1, far1
#define PKA_XP "v8", "v9", "v14", "v15"
#define PKA_XM "v9", "v8", "v15", "v14"
#define PKA_YP "v10", "v11", "v16", "v17"
#define PKA_YM "v11", "v10", "v17", "v16"
#define PKA_ZP "v12", "v13", "v18", "v19"
#define PKA_ZM "v13", "v12", "v19", "v18"
// any direction: per-axis min into v24..v29, max in place in the odd register
#define PKA_XG "v24", "v9", "v27", "v15"
#define PKA_YG "v25", "v11", "v28", "v17"
#define PKA_ZG "v26", "v13", "v29", "v19"

#define PKA_MINMAX                                  \
    "v_min_f32_e32 v24, v8, v9\n\t"                 \
    "v_max_f32_e32 v9, v8, v9\n\t"                  \
    "v_min_f32_e32 v25, v10, v11\n\t"               \
    "v_max_f32_e32 v11, v10, v11\n\t"               \
    "v_min_f32_e32 v26, v12, v13\n\t"               \
    "v_max_f32_e32 v13, v12, v13\n\t"               \
    "v_min_f32_e32 v27, v14, v15\n\t"               \
    "v_max_f32_e32 v15, v14, v15\n\t"               \
    "v_min_f32_e32 v28, v16, v17\n\t"               \
    "v_max_f32_e32 v17, v16, v17\n\t"               \
    "v_min_f32_e32 v29, v18, v19\n\t"               \
    "v_max_f32_e32 v19, v18, v19\n\t"

#define PKA_BODY(PRE, NX0, FX0, NX1, FX1, NY0, FY0, NY1, FY1, NZ0, FZ0, NZ1, FZ1)                      \
    "s_mov_b32 s86, m0\n\t"                                                                             \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                        \
    "s_cbranch_scc1 L_exit_%=\n"                                                                        \
    "L_loop_%=:\n\t"                                                                                    \
    "s_lshl_b32 s87, %[cur], 6\n\t"                                                                     \
    "s_load_dwordx16 s[60:75], %[base], s87\n\t"                                                        \
    "s_waitcnt lgkmcnt(0)\n\t"                                                                          \
    "v_pk_fma_f32 v[8:9], s[60:61], %[A], %[B] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                    \
    "v_pk_fma_f32 v[10:11], s[62:63], %[A], %[C] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"                  \
    "v_pk_fma_f32 v[12:13], s[64:65], %[B], %[C] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                  \
    "v_pk_fma_f32 v[14:15], s[66:67], %[A], %[B] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                  \
    "v_pk_fma_f32 v[16:17], s[68:69], %[A], %[C] op_sel:[0,1,0] op_sel_hi:[1,1,0]\n\t"                  \
    "v_pk_fma_f32 v[18:19], s[70:71], %[B], %[C] op_sel:[0,0,1] op_sel_hi:[1,0,1]\n\t"                  \
    PRE                                                                                                 \
    "v_max3_f32 v20, " NX0 ", " NY0 ", " NZ0 "\n\t"                                                     \
    "v_min3_f32 v21, " FX0 ", " FY0 ", " FZ0 "\n\t"                                                     \
    "v_max3_f32 v22, " NX1 ", " NY1 ", " NZ1 "\n\t"                                                     \
    "v_min3_f32 v23, " FX1 ", " FY1 ", " FZ1 "\n\t"                                                     \
    "v_max_f32_e32 v20, 0, v20\n\t"                                                                     \
    "v_mul_f32_e32 v21, 0x3f800004, v21\n\t"                                                            \
    "v_max_f32_e32 v22, 0, v22\n\t"                                                                     \
    "v_mul_f32_e32 v23, 0x3f800004, v23\n\t"                                                            \
    "v_min_f32_e32 v21, v21, %[tmax]\n\t"                                                               \
    "v_min_f32_e32 v23, v23, %[tmax]\n\t"                                                               \
    "v_cmp_le_f32_e64 s[76:77], v20, v21\n\t"                                                           \
    "v_cmp_le_f32_e64 s[78:79], v22, v23\n\t"                                                           \
    "s_and_b64 s[76:77], s[76:77], %[alive]\n\t"                                                        \
    "s_cbranch_scc0 L_no0_%=\n\t"                                                                       \
    "s_and_b64 s[78:79], s[78:79], %[alive]\n\t"                                                        \
    "s_cbranch_scc0 L_only0_%=\n\t"                                                                     \
    /* both boxes hit by some lane: lanes vote for the child to enter first */                          \
    "v_cmp_le_f32_e32 vcc, v20, v22\n\t"                                                                \
    "s_or_b64 s[82:83], s[76:77], s[78:79]\n\t"                                                         \
    "s_bcnt1_i32_b64 s85, s[82:83]\n\t"                                                                 \
    "s_orn2_b64 s[80:81], vcc, s[78:79]\n\t"                                                            \
    "s_and_b64 s[80:81], s[80:81], s[76:77]\n\t"                                                        \
    "s_bcnt1_i32_b64 s84, s[80:81]\n\t"                                                                 \
    "s_sub_i32 s85, s85, s84\n\t"                                                                       \
    "s_cmp_gt_i32 s85, s84\n\t"                                                                         \
    "s_cselect_b32 s80, s72, s73\n\t"                                                                   \
    "s_cselect_b32 %[cur], s73, s72\n\t"                                                                \
    "s_mov_b32 m0, %[sp]\n\t"                                                                           \
    "s_add_i32 %[sp], %[sp], 1\n\t"                                                                     \
    "v_writelane_b32 %[vs], s80, m0\n"                                                                  \
    "L_next_%=:\n\t"                                                                                    \
    "s_add_i32 %[steps], %[steps], 1\n\t"                                                               \
    "s_cmp_lt_i32 %[cur], 0\n\t"                                                                        \
    "s_cbranch_scc0 L_loop_%=\n\t"                                                                      \
    "s_branch L_exit_%=\n"                                                                              \
    "L_only0_%=:\n\t"                                                                                   \
    "s_mov_b32 %[cur], s72\n\t"                                                                         \
    "s_branch L_next_%=\n"                                                                              \
    "L_no0_%=:\n\t"                                                                                     \
    "s_and_b64 s[78:79], s[78:79], %[alive]\n\t"                                                        \
    "s_cbranch_scc0 L_pop_%=\n\t"                                                                       \
    "s_mov_b32 %[cur], s73\n\t"                                                                         \
    "s_branch L_next_%=\n"                                                                              \
    "L_pop_%=:\n\t"                                                                                     \
    "s_cmp_eq_u32 %[sp], 0\n\t"                                                                         \
    "s_cbranch_scc1 L_done_%=\n\t"                                                                      \
    "s_sub_i32 %[sp], %[sp], 1\n\t"                                                                     \
    "s_nop 0\n\t"                                                                                       \
    "v_readlane_b32 %[cur], %[vs], %[sp]\n\t"                                                           \
    "s_branch L_next_%=\n"                                                                              \
    "L_done_%=:\n\t"                                                                                    \
    "s_add_i32 %[steps], %[steps], 1\n\t"                                                               \
    "s_mov_b32 %[cur], 0x80000000\n"                                                                    \
    "L_exit_%=:\n\t"                                                                                    \
    "s_mov_b32 m0, s86\n\t"

#define PKA_BODY_(...) PKA_BODY(__VA_ARGS__)

#define PKA_CLOBBERS                                                                                         \
    "s60", "s61", "s62", "s63", "s64", "s65", "s66", "s67", "s68", "s69", "s70", "s71", "s72", "s73", "s74",  \
    "s75", "s76", "s77", "s78", "s79", "s80", "s81", "s82", "s83", "s84", "s85", "s86", "s87", "v8", "v9",    \
    "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24",  \
    "v25", "v26", "v27", "v28", "v29", "vcc", "scc"

typedef float pka_v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned long long pka_uniform64(unsigned long long v) {
    unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v);
    unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

#define PKA_DEFINE(NAME, PRE, X, Y, Z)                                                                       \
    __device__ __forceinline__ void NAME(const void* base, unsigned long long alive, pka_v2f A, pka_v2f B,   \
                                         pka_v2f C, float tmax, int& cur, int& sp, int& steps, int& vs) {     \
        /* every scalar operand is wave-uniform by construction; say so where the compiler cannot tell */   \
        cur = __builtin_amdgcn_readfirstlane(cur);                                                           \
        sp = __builtin_amdgcn_readfirstlane(sp);                                                             \
        steps = __builtin_amdgcn_readfirstlane(steps);                                                       \
        alive = pka_uniform64(alive);                                                                        \
        base = (const void*)pka_uniform64((unsigned long long)base);                                         \
        asm volatile(PKA_BODY_(PRE, X, Y, Z)                                                                 \
                     : [cur] "+s"(cur), [sp] "+s"(sp), [steps] "+s"(steps), [vs] "+v"(vs)                     \
                     : [base] "s"(base), [alive] "s"(alive), [A] "v"(A), [B] "v"(B), [C] "v"(C),              \
                       [tmax] "v"(tmax)                                                                       \
                     : PKA_CLOBBERS);                                                                         \
    }

PKA_DEFINE(pka_descend_0, "", PKA_XP, PKA_YP, PKA_ZP)
PKA_DEFINE(pka_descend_1, "", PKA_XM, PKA_YP, PKA_ZP)
PKA_DEFINE(pka_descend_2, "", PKA_XP, PKA_YM, PKA_ZP)
PKA_DEFINE(pka_descend_3, "", PKA_XM, PKA_YM, PKA_ZP)
PKA_DEFINE(pka_descend_4, "", PKA_XP, PKA_YP, PKA_ZM)
PKA_DEFINE(pka_descend_5, "", PKA_XM, PKA_YP, PKA_ZM)
PKA_DEFINE(pka_descend_6, "", PKA_XP, PKA_YM, PKA_ZM)
PKA_DEFINE(pka_descend_7, "", PKA_XM, PKA_YM, PKA_ZM)
PKA_DEFINE(pka_descend_any, PKA_MINMAX, PKA_XG, PKA_YG, PKA_ZG)
