/* msd_siggen.h -- seeded synthetic IQ capture generator (see msd_siggen.c). */
#ifndef MSD_SIGGEN_H
#define MSD_SIGGEN_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MSD_SIGGEN_BLOCK 4096u
enum { MSD_SIGGEN_UC8 = 0, MSD_SIGGEN_SC16 = 1, MSD_SIGGEN_SC16Q11 = 2 };

typedef struct msd_siggen_cfg {
    uint64_t seed;
    uint32_t format;           /* MSD_SIGGEN_* (same numbering as convert.h input_format_t) */
    uint32_t slot_samples;     /* one Mode S frame per this many samples; 0 = none (1200 = 2000/s) */
    uint32_t ac_slot_samples;  /* one Mode A/C reply per this many samples; 0 = none */
    uint32_t noise_q16;        /* I/Q noise sigma, 1/65536 of full scale (1311 = 0.02 FS) */
    uint32_t n_aircraft;       /* address pool size */
    uint32_t flip_permille;    /* frames with one flipped bit */
    uint32_t overlap_permille; /* frames placed on top of the previous one */
    uint32_t reserved;
} msd_siggen_cfg;

/* One block of up to MSD_SIGGEN_BLOCK samples starting at sample block_index*MSD_SIGGEN_BLOCK. */
void msd_siggen_block(const msd_siggen_cfg *cfg, uint64_t block_index, uint32_t nsamples, void *out);
/* nsamples samples starting at first_sample (a multiple of MSD_SIGGEN_BLOCK), on nthreads threads.
 * Returns 0 or a negative errno value. */
int msd_siggen_generate(const msd_siggen_cfg *cfg, uint64_t first_sample, uint64_t nsamples,
                        void *out, unsigned nthreads);
/* k-th address of the pool */
uint32_t msd_siggen_aircraft(const msd_siggen_cfg *cfg, uint32_t k);

#ifdef __cplusplus
}
#endif
#endif
