/* bla_host.h -- private helpers of the host (drop-in) layer: lazy runtime start-up, grow-only device
 * staging buffers, and fail-loudly error handling.  Not part of the reference API. */
#ifndef BLA_HOST_H
#define BLA_HOST_H

#include <stddef.h>
#include "bla.h"   /* build with -I <repo>/include */
#include "matrix.h"

#define BLA_HOST_SLOTS 12

void bla_host_init(void);                              /* bla_init(BLA_DEVICE or 0) on first use; exit(1) on failure */
float* bla_host_buf(int slot, size_t floats);          /* device buffer of >= floats elements, reused across calls */
float* bla_host_up(int slot, const float* h, size_t floats);   /* stage host -> device slot, returns device pointer */
void bla_host_down(float* h, const float* d, size_t floats);   /* device -> host, waits for completion */
/* The reference keeps every channel / kernel plane in its own malloc'd Matrix (model/cifar_unet.c:255-292): up to 256 x 256 = 65,536 planes of
 * 36 bytes for one kernel set.  Each operand crosses the bus as ONE copy: the planes are packed into a grow-only host staging block first
 * (block 0: towards the device, block 1: back), or copied straight from the caller's memory when they happen to lie back to back. */
float* bla_host_pack_block(int which, size_t floats);
float* bla_host_up_planes(int slot, Matrix* planes, int count);           /* `count` equally sized matrices -> one contiguous device buffer */
void bla_host_down_planes(Matrix* planes, int count, const float* d);     /* ... and back; waits for completion */
void bla_host_fail(const char* what, bla_status st);   /* prints what + bla_last_error(), exits 1 */
int bla_host_strict(void);                             /* BLA_STRICT_REFERENCE=1: literal reference behaviour where defined */

#define BLA_TRY(call)                                    \
	do {                                                 \
		bla_status _st = (call);                         \
		if (_st != BLA_OK) bla_host_fail(#call, _st);    \
	} while (0)

#endif
