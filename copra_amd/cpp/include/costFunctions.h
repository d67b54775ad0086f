// costFunctions.h -- source-level drop-in for copra's header of the same name (reference: include/costFunctions.h:22-219).
// Code written against copra includes this name; everything it declares lives in copra/copra.h (the MI355X-native mirror of the
// API: same namespace, class names, members and exception types, bodies that hand the work to the C ABI of include/copra_hip.h).
#pragma once
#include "copra/copra.h"

