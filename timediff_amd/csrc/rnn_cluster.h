// Shared by the cluster-persistent kernels (rnn_cluster.hip: forward; rnn_cluster_bwd.hip: BPTT sweep): the hand-off
// between the workgroups of a cluster goes through L2 as 8-byte {payload, tag} granules, written and read with relaxed
// agent-scope 64-bit atomics (self-validating: no flag, no fence).  tag = (launch epoch << 12) | (step + 1).
#pragma once
#include "rnn_geom.h"

#define TDC_WAVES 4
#define TDC_SPIN_LIMIT (1 << 22)
#define TDC_CUS 256
#define TDC_RESIDENT_MAX_NU 3   // widths of up to this many unit blocks: the whole layer in ONE workgroup (rnn_cluster.hip, td_rnn_resident_kernel)
unsigned *td_async_error_word(bool create);   // api.cpp
int td_cluster_device_cus(void);               // rnn_cluster.hip: compute units of the current device (cached), at most TDC_CUS
// a poll gave up: every lane of the wave stores the same word (system scope: the word is host memory)
#define TDC_RAISE(p) do { if ((p).err) __hip_atomic_store((p).err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)

__device__ __forceinline__ unsigned long long tdc_granule(unsigned payload, unsigned tag) {
    return ((unsigned long long)tag << 32) | payload;
}

// Two neighbouring granules as ONE 16-byte write-through store (buffer_store_dwordx4 sc1): an 8-byte sc1 store is one
// fabric write per lane, so a pair written separately costs twice the writes; each 8-byte half carries its own tag, so the
// halves need not land together.  A buffer store from the builtin, not inline asm: the compiler counts it when it places
// its vmcnt waits (an asm store is invisible to that count, and a wait for an older load then also waits for the store) and
// covers the data-register hazard of stores wider than 8 bytes itself.  `base` must be wave-uniform.
typedef __attribute__((ext_vector_type(4))) unsigned tdc_u32x4;
__device__ __forceinline__ void tdc_store16_sc1(const void *base, unsigned byte_offset, tdc_u32x4 v) {
    __builtin_amdgcn_raw_buffer_store_b128(v, __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, -1, 0x00020000), (int)byte_offset, 0, 16 /* sc1 */);
}
__device__ __forceinline__ void tdc_store_granule_pair(const void *base, unsigned byte_offset, unsigned long long bits, unsigned tag) {
    tdc_store16_sc1(base, byte_offset, tdc_u32x4{(unsigned)bits, tag, (unsigned)(bits >> 32), tag});
}

// The receiving side of a granule pair: ONE 16-byte load that bypasses this CU's L1 (buffer_load_dwordx4 sc1; the compiler
// tracks its completion like any load's).  8-byte accesses run at 0.54-0.70 of the 16-byte rate, and a step's sweep is
// bound by the number of loads in flight.  `bytes` bounds the descriptor: a load beyond it returns zeros, never faults.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t tdc_rsrc(const void *base, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc((void *)base, 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ tdc_u32x4 tdc_load_granule_pair(__amdgpu_buffer_rsrc_t r, unsigned byte_offset) {
    return __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_offset, 0, 16 /* sc1 */);
}

// Workgroup -> (cluster, member).  Workgroups go to the 8 XCDs round-robin by their index, so the natural order
// (member = index % NU) would spread a cluster over all XCDs.  This map puts the NU members of a cluster on ONE XCD:
// index = cluster % 8 + 8 * member + 8 * NU * (cluster / 8), in a grid rounded up to whole groups of 8 clusters whose
// surplus workgroups leave at once.  Speed only (the hand-off is agent-scope either way): measured against the natural
// order, 3-layer forward at 32 / 512 / 2048 sequences 2.29 / 3.74 / 13.97 -> 2.20 / 3.15 / 12.94 ms.
__device__ __forceinline__ bool tdc_block_map(int block, int nclusters, int nu, int *cluster, int *member) {
    const int span = 8 * nu;
    *cluster = 8 * (block / span) + block % 8;
    *member = (block % span) / 8;
    return *cluster < nclusters;
}
static inline int tdc_grid(int nclusters, int nu) { return 8 * nu * ((nclusters + 7) / 8); }

unsigned td_cluster_next_epoch(void);   // rnn_cluster.hip: one counter for every cluster launch of the process (20 bits)

// ---- BPTT sweep of one layer by the cluster (rnn_cluster_bwd.hip) ------------------------------------------------------
struct TdClusterBwdParams {
    const char *gsrc[2];     // gradient w.r.t. this layer's output (fragments), n_src of them are summed
    int n_src;
    const float *save;       // [dir][l][bblk][NU][5][64][16] fp32
    size_t save_dir;         // floats
    char *daf;               // [dir][l][bblk][4][KH] fragments
    size_t daf_dir;          // bytes
    __bf16 *dapl;            // [dir][4][hpad][L * Bp]
    size_t dapl_dir;         // elements
    const char *wt;          // W_hh^T fragments [rb][nterm][KH] (td_pack_weights_T_kernel), direction 0
    size_t wt_dir;           // bytes
    unsigned long long *inbox;   // [dir][bblk][parity][dst c'][src c][wave][lane][2] granules
    float *db_ih[2], *db_hh[2];  // per direction: bias gradients (accumulated): row sums of d a over steps and sequences
    int L, nbb, dirs, out_frags, H;
    int ngroups;                 // clusters per direction; cluster gi works the batch blocks gi, gi + ngroups, ...
    unsigned epoch;
    unsigned *err;               // host word raised when a poll times out (api.cpp: td_async_error_word), or NULL
};

bool td_bptt_cluster_supported(const TdRnnGeom &g, int nbb, int L);
size_t td_bptt_cluster_inbox_bytes(const TdRnnGeom &g, int nbb);
int td_launch_bptt_cluster(int gru, const TdRnnGeom &g, TdClusterBwdParams &p, hipStream_t stream);   // sets p.epoch, p.ngroups
