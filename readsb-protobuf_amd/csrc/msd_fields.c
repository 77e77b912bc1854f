/* msd_fields.c -- host entry points of the header-field decode (msd_fields_impl.h has the code that is
 * shared with the emit kernel). */
#define _DEFAULT_SOURCE /* M_PI */
#include "msd_fields_impl.h"
#include "msd_internal.h"

#include <math.h>
#include <string.h>

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

/* decodeMovementFieldV0 / V2 (mode_s.c:216-259): the midpoint of the speed range a movement code stands for.
 * The constants are doubles and the result is narrowed once, as in the reference's `return` to float. */
static float movement_v0(unsigned m)
{
    if (m >= 125) return 0;
    if (m == 124) return 180;
    if (m >= 109) return (float)(100 + (m - 109 + 0.5) * 5);
    if (m >= 94) return (float)(70 + (m - 94 + 0.5) * 2);
    if (m >= 39) return (float)(15 + (m - 39 + 0.5) * 1);
    if (m >= 13) return (float)(2 + (m - 13 + 0.5) * 0.50);
    if (m >= 9) return (float)(1 + (m - 9 + 0.5) * 0.25);
    if (m >= 2) return (float)(0.125 + (m - 2 + 0.5) * 0.125);
    return 0;
}

static float movement_v2(unsigned m)
{
    if (m >= 125) return 0;
    if (m == 124) return 180;
    if (m >= 109) return (float)(100 + (m - 109 + 0.5) * 5);
    if (m >= 94) return (float)(70 + (m - 94 + 0.5) * 2);
    if (m >= 39) return (float)(15 + (m - 39 + 0.5) * 1);
    if (m >= 13) return (float)(2 + (m - 13 + 0.5) * 0.50);
    if (m >= 9) return (float)(1 + (m - 9 + 0.5) * 0.25);
    if (m >= 3) return (float)(0.125 + (m - 3 + 0.5) * 0.875 / 6);
    if (m >= 2) return (float)(0.125 / 2);
    return 0;
}

void msd_fields_to_float(const msd_fields *f, msd_fields_float *o)
{
    memset(o, 0, sizeof *o);
    o->heading_valid = f->heading_valid;
    o->heading_type = f->heading_type;
    if (f->velocity_valid) { /* ES airborne velocity, subtypes 1 and 2 (mode_s.c:826-843) */
        const int ew = f->ew_vel, ns = f->ns_vel;
        /* the reference hands the double (ns^2 + ew^2 + 0.5) to sqrtf: narrowed to float first */
        o->gs_v0 = o->gs_v2 = o->gs_selected = sqrtf((float)((ns * ns) + (ew * ew) + 0.5));
        o->gs_valid = 1;
        if (o->gs_selected > 0) {
            float ground_track = (float)(atan2(ew, ns) * 180.0 / M_PI);
            if (ground_track < 0)
                ground_track += 360;
            o->heading = ground_track;
            o->heading_type = 1; /* HEADING_GROUND_TRACK */
            o->heading_valid = 1;
        }
    } else if (f->movement) { /* ES surface position (mode_s.c:911-916) */
        o->gs_valid = 1;
        o->gs_selected = o->gs_v0 = movement_v0(f->movement);
        o->gs_v2 = movement_v2(f->movement);
    } else if (f->commb_valid & MSD_COMMB_GS) { /* BDS 5,0 (comm_b.c:575-578) */
        o->gs_valid = 1;
        o->gs_v0 = o->gs_v2 = o->gs_selected = (float)(unsigned)f->gs;
    }
    if (f->heading_valid) {
        if (f->commb_format == 8 || f->commb_format == 9) { /* comm_b.c:485-490,623-628: raw * 90 / 512 (+ 180) */
            float h = (float)((f->heading_raw & 1023u) * 90.0 / 512.0);
            if (f->heading_raw & 1024u)
                h = (float)(h + 180.0);
            o->heading = h;
        } else if (f->metype == 19) {
            o->heading = (float)(f->heading_raw * 360.0 / 1024.0); /* mode_s.c:853 */
        } else {
            o->heading = (float)(f->heading_raw * 360.0 / 128.0); /* surface position, mode_s.c:922 */
        }
    }
    if (f->commb_valid & MSD_COMMB_ROLL) { /* comm_b.c:469-474 */
        const unsigned raw = (unsigned)(f->roll_q & 511);
        float roll = (float)(raw * 45.0 / 256.0);
        if (f->roll_q < 0)
            roll = (float)(roll - 90.0);
        o->roll = roll;
        o->roll_valid = 1;
    }
    if (f->commb_valid & MSD_COMMB_TRACK_RATE) { /* comm_b.c:513-518 */
        const unsigned raw = (unsigned)(f->track_rate_q & 511);
        float r = (float)(raw * 8.0 / 256.0);
        if (f->track_rate_q < 0)
            r = r - 16;
        o->track_rate = r;
        o->track_rate_valid = 1;
    }
    if (f->commb_valid & MSD_COMMB_MACH) { /* comm_b.c:649-651,726-727: a float, widened into the double member */
        const float mach = (float)(f->mach_raw * 2.048 / 512);
        o->mach = mach;
        o->mach_valid = 1;
    }
    if (f->nav_valid & MSD_NAV_QNH) {
        if (f->nav_valid & MSD_NAV_QNH_COMMB)
            o->nav_qnh = (float)(800 + f->nav_qnh_raw * 0.1); /* comm_b.c:323-326 */
        else
            o->nav_qnh = (float)(800.0 + (f->nav_qnh_raw - 1) * 0.8); /* mode_s.c:1212 */
        o->nav_qnh_valid = 1;
    }
    if (f->nav_valid & MSD_NAV_HEADING) {
        if (f->nav_valid & MSD_NAV_HEADING_V2)
            o->nav_heading = (float)(f->nav_heading_raw * 180.0 / 256.0); /* mode_s.c:1219 */
        else
            o->nav_heading = (float)f->nav_heading_raw; /* mode_s.c:1131 */
        o->nav_heading_valid = 1;
    }
}
