// TEST INFRASTRUCTURE (not part of libqgd_hip.so, not declared in include/qgd.h): fault injection into a handle of the library.
// Built into tests/hooks/libqgd_testhooks.so (csrc/Makefile: `make testhooks`); tests/qgd_hooks.py binds it.
//   qgd_comm_debug_fail_at(h, n): the next collective evaluation of the handle fails locally in front of its exchange n-1
//   (n = 1..4: window products, affine parts, [grad | scalars], scalars; 0 = off) -- the library then has to abort its
//   communicator and return QGD_ERR_COMM on this rank while the other ranks run into their own bounded wait.
// It writes one field of the handle (qgd_host.h: comm_fail_at); the library only reads it.
#include "qgd_host.h"

extern "C" int qgd_comm_debug_fail_at(qgd_handle h, int32_t collective)
{
    if (!h || collective < 0 || collective > 4) return QGD_ERR_ARGUMENT;
    h->comm_fail_at = collective;
    return QGD_OK;
}
