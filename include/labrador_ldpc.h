/* labrador_ldpc.h -- source-compatibility shim.
 *
 * C programs written against the reference's header (capi/include/labrador_ldpc.h; e.g. its
 * client capi/examples/example.c:14) include "labrador_ldpc.h".  Putting this directory on the
 * include path instead of the reference's, and linking -llabrador_ldpc_hip -lamdhip64 instead
 * of -llabrador_ldpc, is the whole port: every enum constant, size macro and function of the
 * reference header is declared, under the same name, by labrador_ldpc_hip.h.
 */
#ifndef LABRADOR_LDPC_H
#define LABRADOR_LDPC_H
#include "labrador_ldpc_hip.h"
#endif
