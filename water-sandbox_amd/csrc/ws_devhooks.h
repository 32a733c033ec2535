// ws_devhooks.h -- the one place the library may look at the environment.
#pragma once
// Developer / measurement hooks read from the environment exist only in builds made with -DWS_DEV_HOOKS (the tests'
// tests/libwsfluid_dev.so and tools/ab_build.sh).  The product library reads NO environment variable: what a host may
// choose is a flag or a field of ws_device_cfg (tests/test_no_experiment_switches.py checks the binary's strings).
#ifdef WS_DEV_HOOKS
#include <stdlib.h>
#define WS_DEV_ENV(name) getenv(name)
#else
#define WS_DEV_ENV(name) ((const char *)nullptr)
#endif
