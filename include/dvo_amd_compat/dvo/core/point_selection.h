// Forwarding header for dvo_core/include/dvo/core/point_selection.h: the types live in the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_CORE_POINT_SELECTION_H_
#define DVO_AMD_COMPAT_CORE_POINT_SELECTION_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
