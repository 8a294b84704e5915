// ako_u8_rgb.hip -- the u8 level-0 streaming kernels for RGB pixels (see ako_u8.h)
#define AKO_U8_CH 3
#define AKO_U8_NAME(x) x##_rgb
#include "ako_u8_tu.hip.h"
