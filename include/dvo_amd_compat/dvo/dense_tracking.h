// Forwarding header: put include/dvo_amd_compat in front of dvo_core/include on the include path and
// `#include <dvo/dense_tracking.h>` (dvo_core/include/dvo/dense_tracking.h) resolves to the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_DENSE_TRACKING_H_
#define DVO_AMD_COMPAT_DENSE_TRACKING_H_
#include "../../dvo_amd/dense_tracking.hpp"
#endif
