// two instances (<= 32 decision variables each) per wavefront: see packed_impl.inc
#define COPRA_WAVE_WIDTH 32
#include "packed_impl.inc"
