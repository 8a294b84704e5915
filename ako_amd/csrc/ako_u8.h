// ako_u8.h -- launchers of the u8 level-0 streaming kernels (k_forward_stream_u8 / k_inverse_stream_u8 of
// ako_stream.hip.h), one translation unit per pixel format so that they build in parallel with ako_plan.hip:
// ako_u8_rgba.hip (four bytes per pixel) and ako_u8_rgb.hip (three).  Include after ako_stream.hip.h.
#pragma once

namespace ako
{

void akoLaunchForwardU8_rgba(int kind, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st);
void akoLaunchForwardU8_rgb(int kind, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st);
// level 0 in column groups (k_forward_group_u8; ako_u8_group.hip): workgroups of 8 waves, G.strips = number of groups
void akoLaunchForwardGroupU8_rgba(int kind, const LevelParams& P, const StreamGeom& G, uint32_t blocks, hipStream_t st);
// lean: the lean kernels of ako_u8_lean.hip.h (the caller has checked lean_u8_level(): colour, border rule, width, strips)
// opt: the optimistic fp32 pipeline (false: the exact int16-wrapping kernel, which returns at once unless flagged)
void akoLaunchInverseU8_rgba(int kind, bool opt, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t pairs, hipStream_t st);
void akoLaunchInverseU8_rgb(int kind, bool opt, bool lean, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t pairs, hipStream_t st);

// one- and two-channel images (ako_u8_gray.hip.h): one wave per strip carrying all planes; the inverse is the exact integer
// pipeline (no optimistic launch in front of it); the caller has checked gray_native_level()
void akoLaunchForwardU8_gray(int kind, int channels, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st);
void akoLaunchInverseU8_gray(int kind, int channels, const LevelParams& P, const StreamGeom& G, uint32_t blocks, uint32_t threads, hipStream_t st);

}  // namespace ako
