/* Host build of the S24 encoder (pybmf_amd/csrc/s24.h) for tests/test_s24_format.py: gcc -shared -fPIC. */
#include "../../pybmf_amd/csrc/s24.h"

unsigned s24_code(unsigned m) { return bmf_s24_code(m); }
unsigned s24_encode_pair(const uint32_t* w, uint32_t* idx, uint32_t* val, uint32_t* kept) {
    return bmf_s24_encode_pair((const uint32_t(*)[4])w, (uint32_t(*)[4])idx, (uint32_t(*)[2])val, (uint32_t(*)[4])kept);
}
