/* bla_host.h -- private helpers of the host (drop-in) layer: lazy runtime start-up, grow-only device
 * staging buffers, and fail-loudly error handling.  Not part of the reference API. */
#ifndef BLA_HOST_H
#define BLA_HOST_H

#include <stddef.h>
#include "bla.h"   /* build with -I <repo>/include */

#define BLA_HOST_SLOTS 12

void bla_host_init(void);                              /* bla_init(BLA_DEVICE or 0) on first use; exit(1) on failure */
float* bla_host_buf(int slot, size_t floats);          /* device buffer of >= floats elements, reused across calls */
float* bla_host_up(int slot, const float* h, size_t floats);   /* stage host -> device slot, returns device pointer */
void bla_host_down(float* h, const float* d, size_t floats);   /* device -> host, waits for completion */
void bla_host_fail(const char* what, bla_status st);   /* prints what + bla_last_error(), exits 1 */
int bla_host_strict(void);                             /* BLA_STRICT_REFERENCE=1: literal reference behaviour where defined */

#define BLA_TRY(call)                                    \
	do {                                                 \
		bla_status _st = (call);                         \
		if (_st != BLA_OK) bla_host_fail(#call, _st);    \
	} while (0)

#endif
