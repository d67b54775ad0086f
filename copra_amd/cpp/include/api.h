// api.h -- source-level drop-in for copra's include/api.h:1-47 (symbol visibility macros).
// The mirror is header-only over a C ABI, so user code that decorates its own declarations with these macros gets empty ones.
#pragma once
#ifndef COPRA_DLLAPI
#define COPRA_DLLIMPORT
#define COPRA_DLLEXPORT
#define COPRA_DLLLOCAL
#define COPRA_DLLAPI
#define COPRA_LOCAL
#endif
