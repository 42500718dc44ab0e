// Forwarding header for dvo_core/include/dvo/core/point_selection_predicates.h (included by dvo_slam/src/local_tracker.cpp:24).
// The reference's file is one commented-out block inside empty namespaces: the predicates it once held live in
// point_selection.h (point_selection.h:32-67), mirrored by the MI355X adaptor.
#ifndef DVO_AMD_COMPAT_CORE_POINT_SELECTION_PREDICATES_H_
#define DVO_AMD_COMPAT_CORE_POINT_SELECTION_PREDICATES_H_
#include "../../../dvo_amd/dense_tracking.hpp"
#endif
