/*
 * boundary_check.c -- compile-and-run check of the reference-shaped host boundary
 * (include/modes_hip_readsb.h) from the point of view of a program written against the reference's
 * headers: the converter types are re-declared here the way convert.h:27-45 declares them (this
 * file's own text), MSD_BIND_REFERENCE_CONVERTER is defined, and the exports must then be usable with those types
 * without a cast.  Built with -Werror by tests/test_boundary.py; runs without a GPU.
 */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

/* --- what a readsb translation unit has in scope from its own convert.h --- */
#define MSD_BIND_REFERENCE_CONVERTER 1
struct converter_state;
typedef enum { INPUT_UC8 = 0, INPUT_SC16, INPUT_SC16Q11 } input_format_t;
typedef void (*iq_convert_fn)(void *iq_data, uint16_t *mag_data, unsigned nsamples, struct converter_state *state,
                              double *out_mean_level, double *out_mean_power);
/* --- and the option keys of its own option table (readsb.h:615-617 are enum members whose values only
 * that build knows; any three distinct numbers do here) --- */
enum { OptPlutoNetworkish = 7000, OptIfileName, OptIfileFormat, OptIfileThrottle };

#include "modes_hip_readsb.h"

static int selected, monitored, eof_seen, exit_polled;
static void on_selected(void) { ++selected; }
static void on_monitor(void) { ++monitored; }
static void on_eof(void) { ++eof_seen; }
static int on_should_exit(void) { ++exit_polled; return 0; }

#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) {                                                                \
            printf("FAILED line %d: %s (%s)\n", __LINE__, #cond, msd_ifileLastError()); \
            return 1;                                                                 \
        }                                                                             \
    } while (0)

int main(int argc, char **argv)
{
    /* the factory and the converter with the reference's own types, no casts */
    iq_convert_fn (*factory)(input_format_t, double, int, struct converter_state **) = msd_init_converter;
    void (*cleanup)(struct converter_state *) = msd_cleanup_converter;
    struct converter_state *state = (struct converter_state *)0x1;
    iq_convert_fn fn = factory(INPUT_SC16Q11, 0.0, 1 /* --dcfilter */, &state);
    CHECK(fn == NULL && state == NULL); /* the DC block's constant exp(-2 pi / sample_rate) needs a rate (convert.c:479-482) */
    fn = factory(INPUT_SC16Q11, 2000000.0, 1, &state); /* ... any rate, as the reference's factory takes it */
    if (fn)
        cleanup(state);
    state = (struct converter_state *)0x1;
    fn = factory((input_format_t)9, 2400000.0, 0, &state);
    CHECK(fn == NULL && state == NULL);
    fn = factory(INPUT_SC16Q11, 2400000.0, 1 /* --dcfilter */, &state);
    if (fn) { /* a GPU is present: convert_sc16q11_generic, its DC estimate carried in the state from call to call */
        static int16_t iq16[2 * 4096];
        static uint16_t mag_a[4096], mag_b[4096];
        double level = -1, power = -1;
        for (unsigned i = 0; i < 2 * 4096; ++i)
            iq16[i] = (int16_t)(300 + (int)((i * 2654435761u) >> 22) - 512); /* an offset for the block to find */
        fn(iq16, mag_a, 4096, state, &level, &power);
        CHECK(msd_converter_error(state)[0] == 0 && level > 0 && power > 0);
        fn(iq16, mag_b, 4096, state, NULL, NULL); /* the same samples again: the estimate has moved on */
        CHECK(memcmp(mag_a, mag_b, sizeof mag_a) != 0);
        cleanup(state);
    } else {
        CHECK(state == NULL);
    }
    fn = factory(INPUT_UC8, 2400000.0, 0, &state);
    if (fn) { /* a GPU is present: one block through the converter */
        static uint8_t iq[2 * 4096];
        static uint16_t mag[4096];
        double level = -1, power = -1;
        for (unsigned i = 0; i < sizeof iq; ++i)
            iq[i] = (uint8_t)(i * 37u + 11u);
        fn(iq, mag, 4096, state, &level, &power);
        CHECK(msd_converter_error(state)[0] == 0 && level > 0 && power > 0);
        fn(iq, mag, 4096, state, NULL, NULL); /* convert.c:104-110 */
        CHECK(msd_converter_context(state) != NULL);
    } else {
        CHECK(state == NULL);
    }
    cleanup(state);
    cleanup(NULL);

    /* the demodulators with the reference's shape (demod_2400.h:37-38): void f(struct mag_buf *) */
    {
        void (*demod)(struct msd_mag_buf *) = msd_demodulate2400;
        void (*demod_ac)(struct msd_mag_buf *) = msd_demodulate2400AC;
        static uint16_t silence[326 + 1024];
        struct msd_mag_buf buf;
        memset(&buf, 0, sizeof buf);
        buf.data = silence;
        buf.totalLength = buf.validLength = 326 + 1024;
        buf.overlap = 326;
        CHECK(msd_demod_bind(NULL, 0, NULL, NULL) == 0);
        demod(&buf); /* nothing bound: says so, delivers nothing */
        CHECK(strstr(msd_demod_error(), "no receiver bound") != NULL);
        demod_ac(&buf);
        demod(NULL);
        demod_ac(NULL);
    }

    /* the handler, driven the way sdr.c drives the one it replaces */
    msd_ifile_hooks hooks = {on_should_exit, on_monitor, on_eof, on_selected};
    msd_ifileSetOptionKeys(OptIfileName, OptIfileFormat, OptIfileThrottle, -1);
    msd_ifileSetHooks(&hooks);
    msd_ifileInitConfig();
    CHECK(msd_ifileHandleOption(OptIfileFormat, "SC16"));
    CHECK(!msd_ifileHandleOption(OptIfileFormat, "bogus"));
    CHECK(strstr(msd_ifileLastError(), "not understood") != NULL);
    CHECK(msd_ifileHandleOption(OptIfileThrottle, NULL));
    CHECK(msd_ifileHandleOption(4242, "ignored")); /* some other device's option: accepted, ignored */
    CHECK(msd_ifileHandleOption(MSD_OPT_IFILE_NAME, "not-a-key-any-more")); /* the built-in keys are replaced */
    CHECK(!msd_ifileOpen());                                               /* ... so there is still no file name */
    CHECK(strstr(msd_ifileLastError(), "requires an --ifile argument") != NULL && selected == 0);
    CHECK(msd_ifileHandleOption(OptIfileName, "/nonexistent/capture.bin"));
    CHECK(selected == 1);
    CHECK(!msd_ifileOpen());
    CHECK(strstr(msd_ifileLastError(), "could not open") != NULL);
    if (argc > 1) { /* a real file: opens (and, without a GPU, fails at the device) */
        msd_ifileInitConfig();
        CHECK(msd_ifileHandleOption(OptIfileName, argv[1]));
        if (msd_ifileOpen()) {
            msd_ifileRun();
            CHECK(eof_seen == 1 && monitored > 0 && exit_polled > 0);
        } else {
            CHECK(strstr(msd_ifileLastError(), "msd_create failed") != NULL);
        }
    }
    msd_ifileClose();
    printf("boundary ok\n");
    return 0;
}
