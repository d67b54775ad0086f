// four instances (<= 16 decision variables each) per wavefront: see packed_impl.inc
#define COPRA_WAVE_WIDTH 16
#include "packed_impl.inc"
