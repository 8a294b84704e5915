// ako_u8_rgba.hip -- the u8 level-0 streaming kernels for RGBA pixels (see ako_u8.h)
#define AKO_U8_CH 4
#define AKO_U8_NAME(x) x##_rgba
#include "ako_u8_tu.hip.h"
