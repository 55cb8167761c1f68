// TEST INFRASTRUCTURE: the host tap builder (picha_amd/csrc/taps.cpp) over a sweep of geometries under ASan + UBSan.
#include <cstdio>
#include "../../picha_amd/csrc/taps.h"
int main() {
    long tables = 0;
    for (int filter = 0; filter < 6; ++filter)
        for (float fs : {0.5f, 0.7f, 1.0f, 2.25f})
            for (int src : {1, 2, 3, 7, 64, 97, 640, 3840, 7680})
                for (int dst : {1, 2, 5, 32, 96, 512, 1920, 4000}) {
                    picha_hip::AxisTaps t;
                    const int rc = picha_hip::build_axis_taps(filter, fs, src, dst, t);
                    if (rc == 0) {
                        if ((int)t.left.size() != dst || (int)t.weight.size() != dst * t.taps) { printf("bad table\n"); return 1; }
                        for (int i = 0; i < dst; ++i)
                            if (t.left[i] < 0 || t.left[i] + t.count[i] > src || t.count[i] < 1 || t.count[i] > t.taps) { printf("bad range\n"); return 1; }
                        ++tables;
                    }
                }
    printf("ok: %ld tables\n", tables);
    return 0;
}
