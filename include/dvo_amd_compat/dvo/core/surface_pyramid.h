// Forwarding header for dvo_core/include/dvo/core/surface_pyramid.h (included by dvo_ros/src/camera_dense_tracking.cpp:27):
// dvo::core::SurfacePyramid::convertRawDepthImage / convertRawDepthImageSse live in the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_CORE_SURFACE_PYRAMID_H_
#define DVO_AMD_COMPAT_CORE_SURFACE_PYRAMID_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
