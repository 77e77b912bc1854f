/* msd_pred_impl.h -- the prediction table of a batch on the device (msd_internal.h: MSD_PRED_*): filled by the
 * scan kernel as it finds CRC-clean DF17 / DF11(II=0) tries, read by the resolve kernel.  Device code only. */
#ifndef MSD_PRED_IMPL_H
#define MSD_PRED_IMPL_H

#include "msd_internal.h"

__device__ __forceinline__ unsigned long long msd_pred_key(uint32_t gen, uint32_t addr)
{
    return ((unsigned long long)((gen << 24) | (addr & 0xffffffu))) << 32;
}

/* `addr` has a clean squitter in buffer `buffer`: claim a slot (entries of other generations are vacant) and lower
 * the entry's first buffer.  mode_s.c:717-726 says which messages reach icaoFilterAdd. */
__device__ inline void msd_pred_note(unsigned long long *table, uint32_t gen, uint32_t addr, uint32_t buffer)
{
    uint32_t *count = reinterpret_cast<uint32_t *>(table + MSD_PRED_SLOTS), *list = count + 2;
    const unsigned long long key = msd_pred_key(gen, addr);
    uint32_t h = MSD_PRED_HASH(addr) & (MSD_PRED_SLOTS - 1);
    /* at most one trip round the table: a batch with more clean squitter addresses than the list holds is resolved on
     * the host anyway (the count says so), and a probe must end even if a flood of them filled every slot */
    for (uint32_t probes = 0; probes < MSD_PRED_SLOTS; ++probes) {
        const unsigned long long e = __atomic_load_n(&table[h], __ATOMIC_RELAXED);
        if ((e & 0xffffffff00000000ull) == key) { /* most tries find their aircraft's slot */
            if ((uint32_t)e > buffer)
                atomicMin(&table[h], key | buffer);
            return;
        }
        if ((uint32_t)(e >> 56) != gen) { /* vacant: another generation's, or never used */
            /* the list is full: no more entries (the table stays at most half full), only the count goes past the list's
             * size once, which hands the batch to the host resolver (msd_pred_count) */
            const uint32_t seen = __atomic_load_n(count, __ATOMIC_RELAXED);
            if ((seen >> 24) == gen && (seen & 0xffffffu) >= MSD_PRED_LIST) {
                if ((seen & 0xffffffu) == MSD_PRED_LIST)
                    atomicCAS(count, seen, seen + 1u);
                return;
            }
            if (atomicCAS(&table[h], e, key | buffer) == e) {
                /* the counter cell carries the generation too: the first entry of a batch restarts it */
                uint32_t k;
                for (;;) {
                    const uint32_t cnt = __atomic_load_n(count, __ATOMIC_RELAXED);
                    if ((cnt >> 24) != gen) {
                        if (atomicCAS(count, cnt, (gen << 24) | 1u) == cnt) {
                            k = 0;
                            break;
                        }
                    } else {
                        k = atomicAdd(count, 1u) & 0xffffffu;
                        break;
                    }
                }
                if (k < MSD_PRED_LIST)
                    list[k] = h;
                return;
            }
            --probes;
            continue; /* somebody else took it: look at the slot again */
        }
        h = (h + 1) & (MSD_PRED_SLOTS - 1);
    }
}

/* the first buffer with a clean squitter of addr, or MSD_PRED_NEVER */
__device__ __forceinline__ uint32_t msd_pred_lookup(const unsigned long long *table, uint32_t gen, uint32_t addr)
{
    const unsigned long long key = msd_pred_key(gen, addr);
    uint32_t h = MSD_PRED_HASH(addr) & (MSD_PRED_SLOTS - 1);
    for (uint32_t probes = 0; probes < MSD_PRED_SLOTS; ++probes) { /* (a full table of this generation: one trip round) */
        const unsigned long long e = table[h];
        if ((e & 0xffffffff00000000ull) == key)
            return (uint32_t)e;
        if ((uint32_t)(e >> 56) != gen)
            return MSD_PRED_NEVER;
        h = (h + 1) & (MSD_PRED_SLOTS - 1);
    }
    return MSD_PRED_NEVER;
}

/* entries of the current generation (0 if the batch noted none) */
__device__ __forceinline__ uint32_t msd_pred_count(const unsigned long long *table, uint32_t gen)
{
    const uint32_t cnt = *reinterpret_cast<const uint32_t *>(table + MSD_PRED_SLOTS);
    return (cnt >> 24) == gen ? (cnt & 0xffffffu) : 0u;
}

#endif
