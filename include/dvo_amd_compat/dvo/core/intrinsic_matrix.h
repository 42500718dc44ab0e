// Forwarding header for dvo_core/include/dvo/core/intrinsic_matrix.h: the types live in the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_CORE_INTRINSIC_MATRIX_H_
#define DVO_AMD_COMPAT_CORE_INTRINSIC_MATRIX_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
