/* msd_fields_impl.h -- header fields of an accepted message, one implementation for the emit kernel
 * (device) and the host paths: what decodeModesMessage fills in after its CRC switch without looking
 * into the ME / MB payloads (mode_s.c:557-715), decodeAC13Field / decodeID13Field (mode_s.c:101-183),
 * the Gillham altitude of mode_ac.c:101-163, and decodeModeAMessage (mode_ac.c:168-202).
 * Then the payloads: the extended squitter of DF17/18 (decodeExtendedSquitter and its type decoders,
 * mode_s.c:736-1474) and the Comm-B register inference of DF20/21 (decodeCommB, comm_b.c).
 * SURVEY.md 8(f) rank 1. */
#ifndef MSD_FIELDS_IMPL_H
#define MSD_FIELDS_IMPL_H

#include <math.h>
#include <stdint.h>

#include "modes_hip.h"

#ifdef __HIPCC__
#define MSD_HD __host__ __device__ static inline
#else
#define MSD_HD static inline
#endif

/* bits first..last of the message, numbered from 1 at the most significant bit (mode_s.h:57-102);
 * at most 32 bits that span at most four bytes */
MSD_HD uint32_t msd_field_bits(const uint8_t *msg, int first, int last)
{
    const int fb = (first - 1) >> 3, lb = (last - 1) >> 3;
    uint32_t v = 0;
    for (int i = fb; i <= lb; ++i)
        v = (v << 8) | msg[i];
    v >>= 7 - ((last - 1) & 7);
    return v & (uint32_t)((1ull << (last - first + 1)) - 1ull);
}

/* 13-bit identity code -> four octal digits, hex-coded A4A2A1 B4B2B1 C4C2C1 D4D2D1 (mode_s.c:101-143);
 * the X/M bit (0x0040) is dropped */
MSD_HD uint32_t msd_id13_to_squawk(uint32_t id13)
{
    uint32_t g = 0;
    g |= (id13 & 0x1000u) ? 0x0010u : 0; /* C1 */
    g |= (id13 & 0x0800u) ? 0x1000u : 0; /* A1 */
    g |= (id13 & 0x0400u) ? 0x0020u : 0; /* C2 */
    g |= (id13 & 0x0200u) ? 0x2000u : 0; /* A2 */
    g |= (id13 & 0x0100u) ? 0x0040u : 0; /* C4 */
    g |= (id13 & 0x0080u) ? 0x4000u : 0; /* A4 */
    g |= (id13 & 0x0020u) ? 0x0100u : 0; /* B1 */
    g |= (id13 & 0x0010u) ? 0x0001u : 0; /* D1 / Q */
    g |= (id13 & 0x0008u) ? 0x0200u : 0; /* B2 */
    g |= (id13 & 0x0004u) ? 0x0002u : 0; /* D2 */
    g |= (id13 & 0x0002u) ? 0x0400u : 0; /* B4 */
    g |= (id13 & 0x0001u) ? 0x0004u : 0; /* D4 */
    return g;
}

/* Gillham-coded Mode A value -> Mode C altitude in hundreds of feet, MSD_INVALID_ALTITUDE if it is
 * not one (mode_ac.c:78-163; the lookup there only keeps the twelve code bits of its argument) */
MSD_HD int32_t msd_mode_a_to_c(uint32_t mode_a)
{
    mode_a &= 0x7777u;
    if ((mode_a & 0x0001u) || !(mode_a & 0x00F0u)) /* D1 set is illegal, C1..C4 cannot all be zero */
        return MSD_INVALID_ALTITUDE;
    uint32_t hundreds = 0, five_hundreds = 0;
    if (mode_a & 0x0010u) hundreds ^= 7;   /* C1 */
    if (mode_a & 0x0020u) hundreds ^= 3;   /* C2 */
    if (mode_a & 0x0040u) hundreds ^= 1;   /* C4 */
    if ((hundreds & 5) == 5)
        hundreds ^= 2;                     /* 7 <-> 5 */
    if (hundreds > 5)
        return MSD_INVALID_ALTITUDE;
    if (mode_a & 0x0002u) five_hundreds ^= 0x0FF; /* D2 */
    if (mode_a & 0x0004u) five_hundreds ^= 0x07F; /* D4 */
    if (mode_a & 0x1000u) five_hundreds ^= 0x03F; /* A1 */
    if (mode_a & 0x2000u) five_hundreds ^= 0x01F; /* A2 */
    if (mode_a & 0x4000u) five_hundreds ^= 0x00F; /* A4 */
    if (mode_a & 0x0100u) five_hundreds ^= 0x007; /* B1 */
    if (mode_a & 0x0200u) five_hundreds ^= 0x003; /* B2 */
    if (mode_a & 0x0400u) five_hundreds ^= 0x001; /* B4 */
    if (five_hundreds & 1)
        hundreds = 6 - hundreds;
    return (int32_t)(five_hundreds * 5 + hundreds) - 13;
}

/* 13-bit altitude code (mode_s.c:152-183): feet; metric altitudes are not decoded */
MSD_HD int32_t msd_ac13_altitude(uint32_t ac13, uint8_t *unit)
{
    if (ac13 & 0x0040u) { /* M bit */
        *unit = 1;
        return MSD_INVALID_ALTITUDE;
    }
    *unit = 0;
    if (ac13 & 0x0010u) { /* Q bit: 25 ft steps */
        const int32_t n = (int32_t)(((ac13 & 0x1F80u) >> 2) | ((ac13 & 0x0020u) >> 1) | (ac13 & 0x000Fu));
        return n * 25 - 1000;
    }
    const int32_t n = msd_mode_a_to_c(msd_id13_to_squawk(ac13));
    if (n < -12)
        return MSD_INVALID_ALTITUDE;
    return 100 * n;
}

/* 12-bit altitude code of the airborne position squitter (mode_s.c:187-208): always feet */
MSD_HD int32_t msd_ac12_altitude(uint32_t ac12)
{
    if (ac12 & 0x10u) /* Q bit */
        return (int32_t)(((ac12 & 0x0FE0u) >> 1) | (ac12 & 0x000Fu)) * 25 - 1000;
    const int32_t n = msd_mode_a_to_c(msd_id13_to_squawk(((ac12 & 0x0FC0u) << 1) | (ac12 & 0x003Fu)));
    return n < -12 ? MSD_INVALID_ALTITUDE : 100 * n;
}

/* setIMF (mode_s.c:770-792): the address is not an ICAO address after all */
MSD_HD void msd_fields_set_imf(msd_fields *f)
{
    f->addr |= MSD_NON_ICAO_ADDRESS;
    f->imf = 1;
    if (f->addrtype == 0 || f->addrtype == 1)
        f->addrtype = 4; /* ADSB_ICAO / ADSB_ICAO_NT -> ADSB_OTHER */
    else if (f->addrtype == 3)
        f->addrtype = 6; /* TISB_ICAO -> TISB_TRACKFILE */
    else if (f->addrtype == 2)
        f->addrtype = 5; /* ADSR_ICAO -> ADSR_OTHER */
}

/* ME type 29, target state and status (decodeESTargetStatus, mode_s.c:1058-1249) */
MSD_HD void msd_fields_es_target_state(const uint8_t *me, int check_imf, msd_fields *f)
{
    f->mesub = (uint8_t)msd_field_bits(me, 6, 7); /* two bits of subtype only */
    if (check_imf && msd_field_bits(me, 51, 51))
        msd_fields_set_imf(f);
    if (f->mesub == 0 && !msd_field_bits(me, 11, 11)) { /* the DO-260A layout */
        const uint32_t vsrc = msd_field_bits(me, 8, 9);
        const uint32_t src = vsrc == 1 ? 3u /* MCP */ : (vsrc == 2 ? 2u /* aircraft */ : (vsrc == 3 ? 4u /* FMS */ : 0u));
        f->nav_altitude_source = (uint8_t)src;
        const uint32_t vmode = msd_field_bits(me, 14, 15); /* 1 acquiring, 2 maintaining */
        if (vmode == 1 || vmode == 2) {
            f->nav_valid |= MSD_NAV_MODES;
            f->nav_modes |= src == 4 ? 2u : ((vmode == 2 && src == 2) ? 4u : 1u); /* VNAV / altitude hold / autopilot */
        }
        const int32_t alt = -1000 + 100 * (int32_t)msd_field_bits(me, 16, 25);
        if (src == 3) {
            f->nav_valid |= MSD_NAV_MCP_ALTITUDE;
            f->nav_mcp_altitude = alt;
        } else if (src == 4) {
            f->nav_valid |= MSD_NAV_FMS_ALTITUDE;
            f->nav_fms_altitude = alt;
        }
        const uint32_t hsrc = msd_field_bits(me, 26, 27);
        if (hsrc) {
            f->nav_valid |= MSD_NAV_HEADING;
            f->nav_heading_raw = (uint16_t)msd_field_bits(me, 28, 36);
            f->nav_heading_type = msd_field_bits(me, 37, 37) ? 1 : 4; /* ground track : magnetic or true */
        }
        const uint32_t hmode = msd_field_bits(me, 38, 39);
        if (hmode == 1 || hmode == 2) {
            f->nav_valid |= MSD_NAV_MODES;
            f->nav_modes |= hsrc == 3 ? 16u : 1u; /* LNAV when the FMS steers */
        }
        f->acc_valid |= MSD_ACC_NAC_P | MSD_ACC_NIC_BARO;
        f->nac_p = (uint8_t)msd_field_bits(me, 40, 43);
        f->nic_baro = (uint8_t)msd_field_bits(me, 44, 44);
        f->sil = (uint8_t)msd_field_bits(me, 45, 46);
        f->sil_type = 1; /* unknown */
        const uint32_t tcas = msd_field_bits(me, 52, 53);
        if (tcas)
            f->nav_valid |= MSD_NAV_MODES;
        if (tcas != 1)
            f->nav_modes |= 32u; /* also for 0: "assume TCAS if we had any other modes", without validating them */
        f->emergency_valid = 1;
        f->emergency = (uint8_t)msd_field_bits(me, 54, 56);
    } else if (f->mesub == 1) { /* DO-260B */
        const uint32_t alt_bits = msd_field_bits(me, 10, 20);
        if (alt_bits) {
            if (msd_field_bits(me, 9, 9)) {
                f->nav_valid |= MSD_NAV_FMS_ALTITUDE;
                f->nav_fms_altitude = (int32_t)(alt_bits - 1) * 32;
            } else {
                f->nav_valid |= MSD_NAV_MCP_ALTITUDE;
                f->nav_mcp_altitude = (int32_t)(alt_bits - 1) * 32;
            }
        }
        const uint32_t baro_bits = msd_field_bits(me, 21, 29);
        if (baro_bits) {
            f->nav_valid |= MSD_NAV_QNH;
            f->nav_qnh_raw = (uint16_t)baro_bits;
        }
        if (msd_field_bits(me, 30, 30)) {
            f->nav_valid |= MSD_NAV_HEADING | MSD_NAV_HEADING_V2;
            f->nav_heading_raw = (uint16_t)msd_field_bits(me, 31, 39);
            f->nav_heading_type = 4;
        }
        f->acc_valid |= MSD_ACC_NAC_P | MSD_ACC_NIC_BARO;
        f->nac_p = (uint8_t)msd_field_bits(me, 40, 43);
        f->nic_baro = (uint8_t)msd_field_bits(me, 44, 44);
        f->sil = (uint8_t)msd_field_bits(me, 45, 46);
        f->sil_type = 1;
        if (msd_field_bits(me, 47, 47)) {
            f->nav_valid |= MSD_NAV_MODES;
            f->nav_modes = (uint8_t)((msd_field_bits(me, 48, 48) ? 1u : 0u) | (msd_field_bits(me, 49, 49) ? 2u : 0u) |
                                     (msd_field_bits(me, 50, 50) ? 4u : 0u) | (msd_field_bits(me, 52, 52) ? 8u : 0u) |
                                     (msd_field_bits(me, 53, 53) ? 32u : 0u) | (msd_field_bits(me, 54, 54) ? 16u : 0u));
        }
    }
}

/* ME type 31, aircraft operational status (decodeESOperationalStatus, mode_s.c:1251-1370) */
MSD_HD void msd_fields_es_opstatus(const uint8_t *me, int check_imf, msd_fields *f)
{
#define MSD_OPS_BIT(n, flag) (msd_field_bits(me, (n), (n)) ? (uint32_t)(flag) : 0u)
    f->mesub = (uint8_t)msd_field_bits(me, 6, 8);
    if (check_imf && msd_field_bits(me, 56, 56))
        msd_fields_set_imf(f);
    if (f->mesub > 1)
        return;
    const uint32_t version = msd_field_bits(me, 41, 43), airborne = f->mesub == 0;
    uint32_t ops = MSD_OPS_VALID | (version << 1);
    if (version == 0) {
        if (airborne && msd_field_bits(me, 9, 10) == 0)
            ops |= (msd_field_bits(me, 12, 12) ? 0u : MSD_OPS_CC_ACAS) | MSD_OPS_BIT(13, MSD_OPS_CC_CDTI);
    } else if (version == 1 || version == 2) {
        if (msd_field_bits(me, 25, 26) == 0) {
            ops |= MSD_OPS_BIT(27, MSD_OPS_OM_ACAS_RA) | MSD_OPS_BIT(28, MSD_OPS_OM_IDENT) | MSD_OPS_BIT(29, MSD_OPS_OM_ATC);
            if (version == 2) {
                ops |= MSD_OPS_BIT(30, MSD_OPS_OM_SAF);
                f->acc_valid |= MSD_ACC_SDA;
                f->sda = (uint8_t)msd_field_bits(me, 31, 32);
            }
        }
        const int cc_ok = msd_field_bits(me, 9, 10) == 0 && (version == 2 || msd_field_bits(me, 13, 14) == 0);
        if (cc_ok && airborne) {
            /* bit 11 means "ACAS operational" in version 2 and "not ACAS" before */
            ops |= (version == 2 ? MSD_OPS_BIT(11, MSD_OPS_CC_ACAS) : (msd_field_bits(me, 11, 11) ? 0u : MSD_OPS_CC_ACAS)) |
                   MSD_OPS_BIT(12, version == 2 ? MSD_OPS_CC_1090_IN : MSD_OPS_CC_CDTI) | MSD_OPS_BIT(15, MSD_OPS_CC_ARV) |
                   MSD_OPS_BIT(16, MSD_OPS_CC_TS) | (msd_field_bits(me, 17, 18) << 13);
            if (version == 2)
                ops |= MSD_OPS_BIT(19, MSD_OPS_CC_UAT_IN);
        } else if (cc_ok) {
            ops |= MSD_OPS_BIT(11, MSD_OPS_CC_POA) | MSD_OPS_BIT(12, version == 2 ? MSD_OPS_CC_1090_IN : MSD_OPS_CC_CDTI) |
                   MSD_OPS_BIT(15, MSD_OPS_CC_B2_LOW) | MSD_OPS_CC_LW_VALID | (msd_field_bits(me, 21, 24) << 19);
            if (version == 2) {
                ops |= MSD_OPS_BIT(16, MSD_OPS_CC_UAT_IN);
                f->nac_v_valid = 1;
                f->nac_v = (uint8_t)msd_field_bits(me, 17, 19);
                f->acc_valid |= MSD_ACC_NIC_C;
                f->nic_c = (uint8_t)msd_field_bits(me, 20, 20);
                f->cc_antenna_offset = (uint8_t)msd_field_bits(me, 33, 40);
            }
        }
        f->acc_valid |= MSD_ACC_NIC_A | MSD_ACC_NAC_P;
        f->nic_a = (uint8_t)msd_field_bits(me, 44, 44);
        f->nac_p = (uint8_t)msd_field_bits(me, 45, 48);
        f->sil = (uint8_t)msd_field_bits(me, 51, 52);
        f->sil_type = version == 2 ? (msd_field_bits(me, 55, 55) ? 2 : 3) : 1; /* per sample : per hour; unknown before */
        const uint32_t hrd = msd_field_bits(me, 54, 54) ? 3u : 2u; /* magnetic : true */
        ops |= hrd << 23;
        if (airborne) {
            if (version == 2) {
                f->acc_valid |= MSD_ACC_GVA;
                f->gva = (uint8_t)msd_field_bits(me, 49, 50);
            }
            f->acc_valid |= MSD_ACC_NIC_BARO;
            f->nic_baro = (uint8_t)msd_field_bits(me, 53, 53);
        } else {
            ops |= (msd_field_bits(me, 53, 53) ? hrd : 1u /* ground track */) << 26; /* TAH, DO-260B 2.2.3.2.7.2.12 */
        }
    }
    f->opstatus = ops;
#undef MSD_OPS_BIT
}

/* decodeExtendedSquitter (mode_s.c:1373-1474) and the per-type decoders it calls; me = msg + 4.
 * f already holds the header fields (CF for DF18, airground from CA for DF17). */
MSD_HD void msd_fields_es(const uint8_t *me, uint32_t df, msd_fields *f)
{
    const uint32_t metype = msd_field_bits(me, 1, 5);
    int check_imf = 0;
    f->metype = (uint8_t)metype;
    if (df == 18) {
        switch (f->CF) {
        case 0: f->addrtype = 1; break;                                   /* ADS-B, non-transponder device */
        case 1: f->addrtype = 4; f->addr |= MSD_NON_ICAO_ADDRESS; break;  /* anonymous / vehicle / obstruction */
        case 2: f->source = 5; f->addrtype = 3; check_imf = 1; break;     /* fine TIS-B */
        case 3:                                                           /* coarse TIS-B: only the IMF bit */
            f->source = 5;
            f->addrtype = 3;
            if (msd_field_bits(me, 1, 1))
                msd_fields_set_imf(f);
            return;
        case 5: f->addrtype = 7; f->source = 5; f->addr |= MSD_NON_ICAO_ADDRESS; break; /* TIS-B, non-ICAO */
        case 6: f->addrtype = 2; f->source = 6; check_imf = 1; break;     /* ADS-R */
        default:
            f->addrtype = 9;
            f->addr |= MSD_NON_ICAO_ADDRESS;
            return;
        }
    }
    if (metype >= 1 && metype <= 4) { /* identification and category, mode_s.c:736-766 */
        const char *ais = "@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_ !\"#$%&'()*+,-./0123456789:;<=>?"; /* ais_charset.c */
        f->mesub = (uint8_t)msd_field_bits(me, 6, 8);
        f->callsign_valid = 1;
        for (int i = 0; i < 8; ++i) {
            const char c = ais[msd_field_bits(me, 9 + 6 * i, 14 + 6 * i)];
            f->callsign[i] = c;
            if (!((c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9') || c == ' '))
                f->callsign_valid = 0;
        }
        f->category = (uint8_t)(((0x0Eu - metype) << 4) | f->mesub);
        f->category_valid = 1;
    } else if (metype == 19) { /* airborne velocity, mode_s.c:794-900 */
        f->mesub = (uint8_t)msd_field_bits(me, 6, 8);
        if (f->mesub < 1 || f->mesub > 4)
            return;
        if (check_imf && msd_field_bits(me, 9, 9))
            msd_fields_set_imf(f);
        f->nac_v_valid = 1;
        f->nac_v = (uint8_t)msd_field_bits(me, 11, 13);
        if (f->mesub <= 2) {
            const int32_t ew_raw = (int32_t)msd_field_bits(me, 15, 24), ns_raw = (int32_t)msd_field_bits(me, 26, 35);
            if (ew_raw && ns_raw) {
                const int32_t scale = f->mesub == 2 ? 4 : 1;
                f->ew_vel = (int16_t)((ew_raw - 1) * (msd_field_bits(me, 14, 14) ? -1 : 1) * scale);
                f->ns_vel = (int16_t)((ns_raw - 1) * (msd_field_bits(me, 25, 25) ? -1 : 1) * scale);
                f->velocity_valid = 1;
            }
        } else {
            if (msd_field_bits(me, 14, 14)) {
                f->heading_valid = 1;
                f->heading_raw = (uint16_t)msd_field_bits(me, 15, 24);
                f->heading_type = 4; /* HEADING_MAGNETIC_OR_TRUE */
            }
            const uint32_t airspeed = msd_field_bits(me, 26, 35);
            if (airspeed) {
                const uint32_t speed = (airspeed - 1) * (f->mesub == 4 ? 4u : 1u);
                if (msd_field_bits(me, 25, 25)) {
                    f->tas_valid = 1;
                    f->tas = (uint16_t)speed;
                } else {
                    f->ias_valid = 1;
                    f->ias = (uint16_t)speed;
                }
            }
        }
        const int32_t vert_rate = (int32_t)msd_field_bits(me, 38, 46);
        if (vert_rate) {
            const int32_t rate = (vert_rate - 1) * (msd_field_bits(me, 37, 37) ? -64 : 64);
            if (msd_field_bits(me, 36, 36)) {
                f->baro_rate = (int16_t)rate;
                f->baro_rate_valid = 1;
            } else {
                f->geom_rate = (int16_t)rate;
                f->geom_rate_valid = 1;
            }
        }
        const int32_t raw_delta = (int32_t)msd_field_bits(me, 50, 56);
        if (raw_delta) {
            f->geom_delta_valid = 1;
            f->geom_delta = (int16_t)((raw_delta - 1) * (msd_field_bits(me, 49, 49) ? -25 : 25));
        }
    } else if (metype >= 5 && metype <= 8) { /* surface position, mode_s.c:902-937 */
        f->airground = 1;
        f->cpr_valid = 1;
        f->cpr_type = 0;
        const uint32_t movement = msd_field_bits(me, 6, 12);
        if (movement > 0 && movement < 125)
            f->movement = (uint8_t)movement;
        if (msd_field_bits(me, 13, 13)) {
            f->heading_valid = 1;
            f->heading_raw = (uint16_t)msd_field_bits(me, 14, 20);
            f->heading_type = 5; /* HEADING_TRACK_OR_HEADING */
        }
        if (check_imf && msd_field_bits(me, 21, 21))
            msd_fields_set_imf(f);
        f->cpr_odd = (uint8_t)msd_field_bits(me, 22, 22);
        f->cpr_lat = msd_field_bits(me, 23, 39);
        f->cpr_lon = msd_field_bits(me, 40, 56);
    } else if (metype == 0 || (metype >= 9 && metype <= 18) || (metype >= 20 && metype <= 22)) {
        /* airborne position, mode_s.c:939-1022 */
        switch (msd_field_bits(me, 6, 7)) { /* surveillance status */
        case 0: f->alert_valid = f->spi_valid = 1; f->alert = f->spi = 0; break;
        case 1: case 2: f->alert_valid = 1; f->alert = 1; break;
        default: f->alert_valid = f->spi_valid = 1; f->alert = 0; f->spi = 1; break;
        }
        if (check_imf) {
            if (msd_field_bits(me, 8, 8))
                msd_fields_set_imf(f);
        } else {
            f->nic_b_valid = 1;
            f->nic_b = (uint8_t)msd_field_bits(me, 8, 8);
        }
        const uint32_t ac12 = msd_field_bits(me, 9, 20);
        if (metype != 0) {
            f->cpr_lat = msd_field_bits(me, 23, 39);
            f->cpr_lon = msd_field_bits(me, 40, 56);
            /* a known transmitter fault: altitude 0, longitude 0, type 15, zeros in the latitude LSBs */
            if (!(ac12 == 0 && f->cpr_lon == 0 && (f->cpr_lat & 0x0fffu) == 0 && metype == 15)) {
                f->cpr_valid = 1;
                f->cpr_type = 1;
                f->cpr_odd = (uint8_t)msd_field_bits(me, 22, 22);
            }
        }
        if (ac12 && f->airground != 1) {
            const int32_t alt = msd_ac12_altitude(ac12);
            if (alt != MSD_INVALID_ALTITUDE) {
                if (metype >= 20) {
                    f->altitude_geom = alt;
                    f->altitude_geom_unit = 0;
                    f->altitude_geom_valid = 1;
                } else {
                    f->altitude_baro = alt;
                    f->altitude_baro_unit = 0;
                    f->altitude_baro_valid = 1;
                }
            }
        }
    } else if (metype == 23) { /* test message, mode_s.c:1024-1036 */
        f->mesub = (uint8_t)msd_field_bits(me, 6, 8);
        if (f->mesub == 7) {
            const uint32_t id13 = msd_field_bits(me, 9, 21);
            if (id13) {
                f->squawk_valid = 1;
                f->squawk = (uint16_t)msd_id13_to_squawk(id13);
            }
        }
    } else if (metype == 28) { /* aircraft status, mode_s.c:1038-1057 */
        f->mesub = (uint8_t)msd_field_bits(me, 6, 8);
        if (f->mesub == 1) {
            f->emergency_valid = 1;
            f->emergency = (uint8_t)msd_field_bits(me, 9, 11);
            const uint32_t id13 = msd_field_bits(me, 12, 24);
            if (id13) {
                f->squawk_valid = 1;
                f->squawk = (uint16_t)msd_id13_to_squawk(id13);
            }
            if (check_imf && msd_field_bits(me, 56, 56))
                msd_fields_set_imf(f);
        }
    } else if (metype == 29)
        msd_fields_es_target_state(me, check_imf, f);
    else if (metype == 31)
        msd_fields_es_opstatus(me, check_imf, f);
    /* 24 (surface system status), 30 (operational coordination) and the rest carry nothing the reference decodes */
}

/* ---- Comm-B (comm_b.c:50-744).  DF20/21 do not say which BDS register the MB field answers: every known
 * layout is scored for plausibility, the best one wins, a tie decodes nothing.  All range checks are done on
 * the raw integers; the thresholds are the reference's float comparisons solved for the raw value
 * (tests/test_fields.py re-evaluates the float expressions over every raw value). ---- */
MSD_HD int msd_commb_valid_char(uint32_t six) /* ais_charset: A-Z, 0-9, space */
{
    return (six >= 1 && six <= 26) || (six >= 48 && six <= 57) || six == 32;
}

/* score of layout k (the order of comm_b_decoders, comm_b.c:39-48): 0 empty, 1 BDS 1,0, 2 BDS 2,0, 3 BDS 3,0,
 * 4 BDS 1,7, 5 BDS 4,0, 6 BDS 5,0, 7 BDS 6,0 */
MSD_HD int msd_commb_score(const uint8_t *mb, int k)
{
#define B1(n) msd_field_bits(mb, (n), (n))
    switch (k) {
    case 0: /* comm_b.c:86-98 */
        for (int i = 0; i < 7; ++i)
            if (mb[i])
                return 0;
        return 56;
    case 1: /* :102-122 */
        return (mb[0] == 0x10 && msd_field_bits(mb, 10, 14) == 0) ? 56 : 0;
    case 2: { /* :207-250 */
        if (mb[0] != 0x20)
            return 0;
        int score = 8;
        for (int i = 0; i < 8; ++i) {
            const uint32_t c = msd_field_bits(mb, 9 + 6 * i, 14 + 6 * i);
            if (msd_commb_valid_char(c))
                score += 6;
            else if (c != 0) /* '@' is padding; anything else cannot be a callsign */
                return 0;
        }
        return score;
    }
    case 3: /* :254-268 */
        return mb[0] == 0x30 ? 56 : 0;
    case 4: { /* BDS 1,7, :126-203 */
        if (msd_field_bits(mb, 25, 56) != 0)
            return 0;
        int score = B1(7) ? 1 : -2;
        score -= 2 * (int)(B1(10) + B1(11) + B1(12) + B1(13) + B1(14) + B1(20) + B1(21) + B1(22));
        const uint32_t es = msd_field_bits(mb, 1, 5);
        if (es == 31)
            score += 5 + (int)B1(6);
        else if (es == 0 && !B1(6))
            score += 1;
        else
            score -= 12;
        if (B1(16) && B1(24))
            score += 2 + (int)B1(9);
        else if (!B1(16) && !B1(24) && !B1(9))
            score += 1;
        else
            score -= 6;
        return score;
    }
    case 5: { /* BDS 4,0, :272-434 */
        const uint32_t mcp_v = B1(1), mcp = msd_field_bits(mb, 2, 13), fms_v = B1(14), fms = msd_field_bits(mb, 15, 26);
        const uint32_t baro_v = B1(27), baro = msd_field_bits(mb, 28, 39), mode_v = B1(48), mode = msd_field_bits(mb, 49, 51);
        const uint32_t src_v = B1(54), src = msd_field_bits(mb, 55, 56);
        if (!mcp_v && !fms_v && !baro_v && !mode_v && !src_v)
            return 0;
        if (msd_field_bits(mb, 40, 47) || msd_field_bits(mb, 52, 53))
            return 0;
        int score = 0;
        const uint32_t alt[2] = {mcp * 16, fms * 16}, altv[2] = {mcp_v, fms_v};
        for (int i = 0; i < 2; ++i) {
            if (altv[i] && alt[i]) {
                if (alt[i] < 1000 || alt[i] > 50000)
                    return 0;
                score += 13;
            } else if (!altv[i] && !alt[i]) {
                score += 1;
            } else {
                return 0;
            }
        }
        if (baro_v && baro) { /* 900 <= 800 + raw * 0.1 <= 1100 */
            if (baro < 1000 || baro > 3000)
                return 0;
            score += 13;
        } else if (!baro_v && !baro) {
            score += 1;
        } else {
            return 0;
        }
        if (mode_v)
            score += 4;
        else if (!mode)
            score += 1;
        else
            return 0;
        if (src_v)
            score += 3;
        else if (!src)
            score += 1;
        else
            return 0;
        if (mcp_v && fms_v && alt[0] != alt[1])
            score -= 4;
        for (int i = 0; i < 2; ++i)
            if (altv[i]) {
                const uint32_t rem = alt[i] % 500;
                if (!(rem < 16 || rem > 484))
                    score -= 4; /* selected altitudes are multiples of 500 ft */
            }
        return score;
    }
    case 6: { /* BDS 5,0, :438-592 */
        const uint32_t roll_v = B1(1), roll_s = B1(2), roll = msd_field_bits(mb, 3, 11);
        const uint32_t trk_v = B1(12), gs_v = B1(24), gs = msd_field_bits(mb, 25, 34) * 2;
        const uint32_t tr_v = B1(35), tr_s = B1(36), tr = msd_field_bits(mb, 37, 45);
        const uint32_t tas_v = B1(46), tas = msd_field_bits(mb, 47, 56) * 2;
        if (!roll_v || !trk_v || !gs_v || !tas_v)
            return 0;
        /* -40 <= roll * 45/256 (- 90) < 40 */
        if (roll_s ? roll < 285 : roll > 227)
            return 0;
        int score = 11 + 12;
        if (gs == 0 || gs < 50 || gs > 700) /* valid with raw 0 is rejected as well (:496-508) */
            return 0;
        score += 11;
        const int trq = (int)tr - (tr_s ? 512 : 0); /* x 1/32 degrees per second */
        if (tr_v) {
            if (trq < -320 || trq > 320)
                return 0;
            score += 11;
        } else if (tr == 0 && !tr_s) {
            score += 1;
        } else {
            return 0;
        }
        if (tas == 0 || tas < 50 || tas > 700)
            return 0;
        score += 11;
        /* (the ground speed / airspeed consistency check compares the two valid bits and never fires, :542-548) */
        if (tr_v) { /* turn rate a coordinated turn at this bank and speed would have, :550-557 */
            const float rollf = (float)((double)roll * 45.0 / 256.0 - (roll_s ? 90.0 : 0.0));
            const float ratef = (float)trq / 32.0f;
            const double turn_rate = 68625 * tan(rollf * 3.14159265358979323846 / 180.0) / (tas * 20 * 3.14159265358979323846);
            if (fabs(turn_rate - ratef) > 2.0)
                score -= 6;
        }
        return score;
    }
    default: { /* BDS 6,0, :596-744 */
        const uint32_t hdg_v = B1(1), ias_v = B1(13), ias = msd_field_bits(mb, 14, 23), mach_v = B1(24), mach = msd_field_bits(mb, 25, 34);
        const uint32_t br_v = B1(35), br_s = B1(36), br = msd_field_bits(mb, 37, 45);
        const uint32_t ir_v = B1(46), ir_s = B1(47), ir = msd_field_bits(mb, 48, 56);
        if (!hdg_v || !ias_v || !mach_v || (!br_v && !ir_v))
            return 0;
        int score = 12;
        if (ias < 50 || ias > 700)
            return 0;
        score += 11;
        if (mach < 25 || mach > 225) /* 0.1 <= raw * 2.048 / 512 <= 0.9 */
            return 0;
        score += 11;
        const int baro_rate = (int)br * 32 - (br_s ? 16384 : 0), inertial_rate = (int)ir * 32 - (ir_s ? 16384 : 0);
        if (br_v) {
            if (baro_rate < -6000 || baro_rate > 6000)
                return 0;
            score += 11;
        } else if (br == 0) { /* the sign bit is not looked at here (:684) */
            score += 1;
        } else {
            return 0;
        }
        if (ir_v) {
            if (inertial_rate < -6000 || inertial_rate > 6000)
                return 0;
            score += 11;
        } else if (ir == 0) {
            score += 1;
        } else {
            return 0;
        }
        if (br_v && ir_v && (baro_rate > inertial_rate ? baro_rate - inertial_rate : inertial_rate - baro_rate) > 2000)
            score -= 12;
        return score;
    }
    }
#undef B1
}

MSD_HD void msd_fields_commb(const uint8_t *mb, msd_fields *f)
{
    f->commb_format = 0;
    /* "If DR or UM are set, this message is probably noise" (comm_b.c:53-58) -- but UM is only extracted after
     * the MB field (mode_s.c:669 before :705), so it still is zero when the reference looks: DR alone decides.
     * (DF20/21 are never bit-corrected either.) */
    if (f->DR != 0)
        return;
    int best = 0, best_k = -1, ties = 0;
    for (int k = 0; k < 8; ++k) {
        const int sc = msd_commb_score(mb, k);
        if (sc > best) {
            best = sc;
            best_k = k;
            ties = 0;
        } else if (sc == best) {
            ties = 1;
        }
    }
    if (best_k < 0)
        return;
    if (ties) {
        f->commb_format = 1;
        return;
    }
#define B1(n) msd_field_bits(mb, (n), (n))
    switch (best_k) {
    case 0: f->commb_format = 2; break;
    case 1: f->commb_format = 3; break;
    case 4: f->commb_format = 4; break;
    case 3: f->commb_format = 6; break;
    case 2: {
        const char *ais = "@ABCDEFGHIJKLMNOPQRSTUVWXYZ[\\]^_ !\"#$%&'()*+,-./0123456789:;<=>?";
        f->commb_format = 5;
        int valid = 1;
        for (int i = 0; i < 8; ++i)
            if (msd_field_bits(mb, 9 + 6 * i, 14 + 6 * i) == 0)
                valid = 0; /* padding: a BDS 2,0 all right, but no callsign to use */
        if (valid) {
            for (int i = 0; i < 8; ++i)
                f->callsign[i] = ais[msd_field_bits(mb, 9 + 6 * i, 14 + 6 * i)];
            f->callsign_valid = 1;
        }
        break;
    }
    case 5:
        f->commb_format = 7;
        if (B1(1)) {
            f->nav_valid |= MSD_NAV_MCP_ALTITUDE;
            f->nav_mcp_altitude = (int32_t)msd_field_bits(mb, 2, 13) * 16;
        }
        if (B1(14)) {
            f->nav_valid |= MSD_NAV_FMS_ALTITUDE;
            f->nav_fms_altitude = (int32_t)msd_field_bits(mb, 15, 26) * 16;
        }
        if (B1(27)) {
            f->nav_valid |= MSD_NAV_QNH | MSD_NAV_QNH_COMMB;
            f->nav_qnh_raw = (uint16_t)msd_field_bits(mb, 28, 39);
        }
        if (B1(48)) {
            const uint32_t m = msd_field_bits(mb, 49, 51);
            f->nav_valid |= MSD_NAV_MODES;
            f->nav_modes = (uint8_t)(((m & 4u) ? 2u : 0u) | ((m & 2u) ? 4u : 0u) | ((m & 1u) ? 8u : 0u)); /* VNAV, hold, approach */
        }
        f->nav_altitude_source = B1(54) ? (uint8_t)(1u + msd_field_bits(mb, 55, 56)) : 0; /* unknown, aircraft, MCP, FMS */
        break;
    case 6:
        f->commb_format = 8;
        f->commb_valid |= MSD_COMMB_ROLL | MSD_COMMB_GS;
        f->roll_q = (int16_t)((int)msd_field_bits(mb, 3, 11) - (B1(2) ? 512 : 0));
        f->heading_valid = 1;
        f->heading_raw = (uint16_t)(msd_field_bits(mb, 14, 23) + (B1(13) ? 1024u : 0u)); /* + 180 degrees */
        f->heading_type = 1; /* ground track */
        f->gs = (uint16_t)(msd_field_bits(mb, 25, 34) * 2);
        if (B1(35)) {
            f->commb_valid |= MSD_COMMB_TRACK_RATE;
            f->track_rate_q = (int16_t)((int)msd_field_bits(mb, 37, 45) - (B1(36) ? 512 : 0));
        }
        f->tas_valid = 1;
        f->tas = (uint16_t)(msd_field_bits(mb, 47, 56) * 2);
        break;
    default:
        f->commb_format = 9;
        f->heading_valid = 1;
        f->heading_raw = (uint16_t)(msd_field_bits(mb, 3, 12) + (B1(2) ? 1024u : 0u));
        f->heading_type = 3; /* magnetic */
        f->ias_valid = 1;
        f->ias = (uint16_t)msd_field_bits(mb, 14, 23);
        f->commb_valid |= MSD_COMMB_MACH;
        f->mach_raw = (uint16_t)msd_field_bits(mb, 25, 34);
        if (B1(35)) {
            f->baro_rate_valid = 1;
            f->baro_rate = (int16_t)((int)msd_field_bits(mb, 37, 45) * 32 - (B1(36) ? 16384 : 0));
        }
        if (B1(46)) { /* INS-derived: a "geometric" rate like elsewhere */
            f->geom_rate_valid = 1;
            f->geom_rate = (int16_t)((int)msd_field_bits(mb, 48, 56) * 32 - (B1(47) ? 16384 : 0));
        }
        break;
    }
#undef B1
}

/* a Mode S message (msgtype 0..31, corrected bytes); addr = msd_message.addr */
MSD_HD void msd_fields_mode_s(const uint8_t *msg, uint32_t df, uint32_t addr, msd_fields *f)
{
    uint8_t *z = (uint8_t *)f;
    for (unsigned i = 0; i < sizeof *f; ++i)
        z[i] = 0;
    f->addr = addr;
    f->source = df == 11 ? 4 : ((df == 17 || df == 18) ? 7 : 3); /* the CRC switch, mode_s.c:447-551 */
    if (df == 0 || df == 4 || df == 16 || df == 20) { /* AC, mode_s.c:565-572 */
        f->AC = (uint16_t)msd_field_bits(msg, 20, 32);
        if (f->AC) {
            f->altitude_baro = msd_ac13_altitude(f->AC, &f->altitude_baro_unit);
            f->altitude_baro_valid = f->altitude_baro != MSD_INVALID_ALTITUDE;
        }
    }
    if (df == 11 || df == 17) { /* CA, mode_s.c:577-597 */
        f->CA = (uint8_t)msd_field_bits(msg, 6, 8);
        if (f->CA == 4)
            f->airground = 1;
        else if (f->CA == 5)
            f->airground = 2;
        else if (f->CA == 0 || f->CA == 6 || f->CA == 7)
            f->airground = 3;
    }
    if (df == 0)
        f->CC = (uint8_t)msd_field_bits(msg, 7, 7);
    if (df == 18)
        f->CF = (uint8_t)msd_field_bits(msg, 6, 8);
    if (df == 4 || df == 5 || df == 20 || df == 21) { /* DR, FS, UM: mode_s.c:610-650,703-705 */
        f->DR = (uint8_t)msd_field_bits(msg, 9, 13);
        f->UM = (uint8_t)msd_field_bits(msg, 14, 19);
        f->FS = (uint8_t)msd_field_bits(msg, 6, 8);
        f->alert_valid = f->spi_valid = f->FS <= 5;
        f->alert = f->FS == 2 || f->FS == 3 || f->FS == 4;
        f->spi = f->FS == 4 || f->FS == 5;
        if (f->FS <= 5)
            f->airground = (f->FS == 1 || f->FS == 3) ? 1 : 3;
    }
    if (df == 5 || df == 21) { /* ID, mode_s.c:653-660 */
        f->ID = (uint16_t)msd_field_bits(msg, 20, 32);
        if (f->ID) {
            f->squawk = (uint16_t)msd_id13_to_squawk(f->ID);
            f->squawk_valid = 1;
        }
    }
    if (df >= 24) { /* KE, ND: mode_s.c:663-665,692-694 */
        f->KE = (uint8_t)msd_field_bits(msg, 4, 4);
        f->ND = (uint8_t)msd_field_bits(msg, 5, 8);
    }
    if (df == 0 || df == 16) { /* RI, SL, VS: mode_s.c:697-716 */
        f->RI = (uint8_t)msd_field_bits(msg, 14, 17);
        f->SL = (uint8_t)msd_field_bits(msg, 9, 11);
        f->VS = (uint8_t)msd_field_bits(msg, 6, 6);
        f->airground = f->VS ? 1 : 3;
    }
    if (df == 17 || df == 18) /* ME, mode_s.c:678-682 */
        msd_fields_es(msg + 4, df, f);
    if (df == 20 || df == 21) /* MB, mode_s.c:666-670 */
        msd_fields_commb(msg + 4, f);
}

/* a Mode A/C reply (mode_ac.c:168-202).  `carry` is the state demodulate2400AC's message record is in
 * when the reply is decoded: it is cleared once per buffer only (demod_2400.c:523-528), so a reply
 * that carries no altitude inherits altitude_baro / _valid / _unit from the last one of the same
 * buffer that did.  NULL = first reply of a buffer. */
MSD_HD void msd_fields_mode_ac(uint32_t mode_a, const msd_fields *carry, msd_fields *f)
{
    uint8_t *z = (uint8_t *)f;
    for (unsigned i = 0; i < sizeof *f; ++i)
        z[i] = 0;
    f->addr = (mode_a & 0x0000FF7Fu) | MSD_NON_ICAO_ADDRESS;
    f->source = 1;   /* SOURCE_MODE_AC */
    f->addrtype = 8; /* ADDR_MODE_A */
    if (carry) {
        f->altitude_baro = carry->altitude_baro;
        f->altitude_baro_valid = carry->altitude_baro_valid;
        f->altitude_baro_unit = carry->altitude_baro_unit;
    }
    f->squawk = (uint16_t)(mode_a & 0x7777u);
    f->squawk_valid = 1;
    f->spi = (mode_a & 0x0080u) ? 1 : 0;
    f->spi_valid = 1;
    if (!f->spi) {
        const int32_t c = msd_mode_a_to_c(mode_a);
        if (c != MSD_INVALID_ALTITUDE) {
            f->altitude_baro = c * 100;
            f->altitude_baro_unit = 0;
            f->altitude_baro_valid = 1;
        }
    }
}

#endif
