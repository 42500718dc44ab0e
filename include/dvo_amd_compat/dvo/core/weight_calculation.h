// Forwarding header for dvo_core/include/dvo/core/weight_calculation.h as far as DenseTracker::Config needs it
// (dvo::core::ScaleEstimators::enum_t / InfluenceFunctions::enum_t + str(), dense_tracking.h:27,56-60).  The estimators
// themselves are never reached by match() and are not mirrored.
#ifndef DVO_AMD_COMPAT_CORE_WEIGHT_CALCULATION_H_
#define DVO_AMD_COMPAT_CORE_WEIGHT_CALCULATION_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
