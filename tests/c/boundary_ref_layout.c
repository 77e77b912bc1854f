/* Pins `struct msd_mag_buf` and the boundary's enums on the TEXT of the reference's own headers (compiled with
 * -I/root/reference in the build container only; the headers are not copied): modes_hip_readsb.h first, so that it
 * declares its own struct, then the reference's fifo.h / convert.h / demod_2400.h beside it.  Every one of the eleven
 * fields of fifo.h:57-73 at the same offset with the same size, the same sizeof, the same flag and format values. */
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#include "modes_hip_readsb.h" /* no MSD_BIND_REFERENCE_* macro: the library's own declarations, whatever else is in scope */
#include "fifo.h"
#include "convert.h"
#include "demod_2400.h"

#define SAME_FIELD(f)                                                                                                   \
    _Static_assert(offsetof(struct mag_buf, f) == offsetof(struct msd_mag_buf, f), "offset of " #f);                    \
    _Static_assert(sizeof(((struct mag_buf *)0)->f) == sizeof(((struct msd_mag_buf *)0)->f), "size of " #f)

SAME_FIELD(data);
SAME_FIELD(totalLength);
SAME_FIELD(validLength);
SAME_FIELD(overlap);
SAME_FIELD(sampleTimestamp);
SAME_FIELD(sysTimestamp);
SAME_FIELD(flags);
SAME_FIELD(mean_level);
SAME_FIELD(mean_power);
SAME_FIELD(dropped);
SAME_FIELD(next);
_Static_assert(sizeof(struct mag_buf) == sizeof(struct msd_mag_buf), "sizeof(struct mag_buf)");
_Static_assert(_Alignof(struct mag_buf) == _Alignof(struct msd_mag_buf), "alignment of struct mag_buf");
_Static_assert((int)MAGBUF_DISCONTINUOUS == (int)MSD_MAGBUF_DISCONTINUOUS, "MAGBUF_DISCONTINUOUS");
_Static_assert(sizeof(mag_buf_flags) == sizeof(msd_mag_buf_flags), "mag_buf_flags");
_Static_assert((int)INPUT_UC8 == (int)MSD_INPUT_UC8 && (int)INPUT_SC16 == (int)MSD_INPUT_SC16 && (int)INPUT_SC16Q11 == (int)MSD_INPUT_SC16Q11,
               "input_format_t values");
_Static_assert(sizeof(input_format_t) == sizeof(msd_input_format_t), "input_format_t");
_Static_assert((int)INPUT_UC8 == MSD_FMT_UC8 && (int)INPUT_SC16 == MSD_FMT_SC16 && (int)INPUT_SC16Q11 == MSD_FMT_SC16Q11,
               "the C-ABI's format codes are the reference's");
/* the thresholds the C-ABI documents (msd_config.preamble_threshold) are the reference's */
_Static_assert(PREAMBLE_THRESHOLD_DEFAULT == 58 && PREAMBLE_THRESHOLD_MIN == 40 && PREAMBLE_THRESHOLD_MAX == 400 && PREAMBLE_THRESHOLD_PIZERO == 75,
               "demod_2400.h thresholds");

int main(void)
{
    printf("layout ok: sizeof(struct mag_buf) = %zu\n", sizeof(struct mag_buf));
    return 0;
}
