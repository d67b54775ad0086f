// debugUtils.h -- source-level drop-in for copra's include/debugUtils.h:15-42: the exception helpers user-written costs and
// constraints throw with (same message prefix "In file ... (line ...): [In function: ...]", same exception types) and the
// deletion warning of LMPC::checkDeleteCostsAndConstraints.
#pragma once
#include "api.h"
#include "copra/copra.h"
#include <cstdio>

#ifndef DOMAIN_ERROR_EXCEPTION
#define DOMAIN_ERROR_EXCEPTION(MESSAGE) COPRA_DOMAIN_ERROR(MESSAGE)
#define RUNTIME_ERROR_EXCEPTION(MESSAGE) COPRA_RUNTIME_ERROR(MESSAGE)
#ifdef NDEBUG
#define CONSTRAINT_DELETION_WARN(warn, format, ...) (void)warn
#else
#define CONSTRAINT_DELETION_WARN(warn, format, ...) \
    do {                                            \
        if (warn) std::fprintf(stderr, format, __VA_ARGS__); \
    } while (0)
#endif
#endif
