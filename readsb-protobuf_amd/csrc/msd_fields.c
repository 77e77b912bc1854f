/* msd_fields.c -- host entry points of the header-field decode (msd_fields_impl.h has the code that is
 * shared with the emit kernel). */
#include "msd_fields_impl.h"
#include "msd_internal.h"

void msd_decode_fields(const msd_message *mm, const msd_fields *carry, msd_fields *out)
{
    if (mm->msgtype == 32)
        msd_fields_mode_ac(((uint32_t)mm->msg[0] << 8) | mm->msg[1], carry, out);
    else
        msd_fields_mode_s(mm->msg, mm->msgtype, mm->addr, out);
}

/* a whole batch on the host (the paths that resolve on host threads): msgs[i] belongs to buffer[i] */
void msd_fields_batch(const void *msgs_base, size_t msg_stride, const uint32_t *buffer, uint64_t n, msd_fields *out)
{
    const msd_fields *carry = NULL;
    uint32_t carry_buffer = 0;
    for (uint64_t i = 0; i < n; ++i) {
        const msd_message *mm = (const msd_message *)((const char *)msgs_base + i * msg_stride);
        if (mm->msgtype == 32) {
            if (carry && carry_buffer != buffer[i])
                carry = NULL; /* demod_2400.c:523-528: the record is cleared once per buffer */
            msd_decode_fields(mm, carry, &out[i]);
            carry = &out[i];
            carry_buffer = buffer[i];
        } else {
            msd_decode_fields(mm, NULL, &out[i]);
        }
    }
}
