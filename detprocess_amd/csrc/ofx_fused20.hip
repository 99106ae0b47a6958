// ofx_fused20.hip -- the register-resident kernel of ofx_fused25.hip built for 20000-sample traces
// (16 ms at 1.25 MHz): M = 10000 = 16 x 25 x 25 packed complex points, a 16-point first stage
// (the radix-2 network of ofx_fft_regs.h), 200 working threads per 256-thread workgroup, three
// full rounds of first-stage transforms and a partial one of 25 virtual threads (wave 0).
#define OFX25_R1 16
#include "ofx_fused25.hip"
