#include <stdint.h>
/* sequential z' = fl(fl(z*b) + t), returns all z */
void chain(const float *t, float *zout, uint64_t n, float z, float b) {
    for (uint64_t i = 0; i < n; i++) { volatile float p = z * b; z = t[i] + p; zout[i] = z; }
}
