// Forwarding header for dvo_core/include/dvo/core/rgbd_image.h: the types live in the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_CORE_RGBD_IMAGE_H_
#define DVO_AMD_COMPAT_CORE_RGBD_IMAGE_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
