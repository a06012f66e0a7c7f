/* Exercises the error behaviour of the drop-in host API without needing a device:
 *   argv[1] = "mul"  -> matrix_multiply on non-conformable operands: reference message on stdout, exit 1
 *   argv[1] = "had"  -> matrix_multiply_elementwise on mismatched shapes: reference message, exit 1
 *   argv[1] = "dev"  -> a compute call with no GPU present must fail loudly (stderr, exit 1), never fall back */
#include "matrix.h"
#include <stdio.h>
#include <string.h>

int main(int argc, char** argv) {
	float d[6] = {1, 2, 3, 4, 5, 6};
	struct Matrix a = {2, 3, d}, b = {2, 3, d}, c = {3, 2, d};
	if (argc < 2) return 2;
	if (!strcmp(argv[1], "mul")) { matrix_multiply(a, b); }
	if (!strcmp(argv[1], "had")) { matrix_multiply_elementwise(&a, &c); }
	if (!strcmp(argv[1], "dev")) { matrix_scale(&a, 2.0f); printf("scaled: %g\n", d[0]); }
	return 0;
}
