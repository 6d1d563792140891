// prints the device's wall-clock (wall_clock64) rate and core clock, in kHz
#include <hip/hip_runtime.h>
#include <cstdio>
int main() {
    int w = 0, c = 0;
    hipDeviceGetAttribute(&w, hipDeviceAttributeWallClockRate, 0);
    hipDeviceGetAttribute(&c, hipDeviceAttributeClockRate, 0);
    printf("wall clock %d kHz, core clock %d kHz\n", w, c);
    return 0;
}
