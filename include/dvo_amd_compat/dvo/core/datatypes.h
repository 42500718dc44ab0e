// Forwarding header for dvo_core/include/dvo/core/datatypes.h (AffineTransformd, Matrix6d, Vector6d ...).
#ifndef DVO_AMD_COMPAT_CORE_DATATYPES_H_
#define DVO_AMD_COMPAT_CORE_DATATYPES_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
