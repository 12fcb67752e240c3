// C wrapper around the REFERENCE's own QMX codec, compiled from the source where it lies
// (/root/reference/qmx_codec.hpp -- self-contained, needs only SSE4.1). Output goes to
// oracle/_ref/libqmx_ref.so (git-ignored, travels to the GPU box as a prebuilt file).
// Test infrastructure only: used to pin oracle.cpp's QMX decoder and the product's QMX
// encoder against real reference bytes. No reference source is copied into this repo.
#include <stddef.h>
#include <stdint.h>
#include REFERENCE_QMX_HEADER

extern "C" size_t ref_qmx_encode(uint8_t* dst, const uint32_t* src) {
    static thread_local QMX::codec<128> c;
    return c.encode(dst, src);
}
// `to` must hold 128 + 512 values (qmx_block::overflow, block_codecs.hpp:319)
extern "C" void ref_qmx_decode(uint32_t* to, const uint8_t* src, size_t len) {
    static thread_local QMX::codec<128> c;
    c.decode(to, src, len);
}
