// Force-included in front of the UNMODIFIED reference src/sbc_decoder.cpp for the efx_ref_sbc
// harness only: gives the two coefficient tables (namespace-scope `const`, hence internal
// linkage in C++) external linkage so the harness can dump them.  TEST INFRASTRUCTURE.
#include <stdint.h>
extern const uint32_t SBC_syn_8[128];
extern const uint32_t SBC_proto_8[80];
