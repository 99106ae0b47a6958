// ofx_fused12.hip -- the register-resident kernel of ofx_fused25.hip built for 12500-sample traces
// (10 ms at 1.25 MHz, the other trace length of the reference's examples:
// /root/reference/examples/filterdata/filter_data_generation.ipynb, trace_length_msec = 10):
// M = 6250 = 10 x 25 x 25 packed complex points, 125 working threads per 128-thread workgroup,
// five full rounds of 10-point transforms in F1 / I1, four workgroups per CU.
#define OFX25_R1 10
#include "ofx_fused25.hip"
