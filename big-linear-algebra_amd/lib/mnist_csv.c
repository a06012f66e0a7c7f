/* mnist_csv.c -- the reference's legacy streaming MNIST reader (lib/mnist_csv.c), restated.  One row per call, straight from the
 * FILE*; values are at most three characters in the reference's files (0..255), the token buffer here is wider and bounded. */
#include "mnist_csv.h"
#include <stdlib.h>

enum { ROW_VALUES = 785, SIDE = 28 };

int get_next_data(struct MnistCSV* csv) {
	if (feof(csv->file)) {                        /* lib/mnist_csv.c:7-10 */
		printf("CSV file is empty\n");
		return 1;
	}
	char token[32];
	int have = 0, filled = 0;
	while (filled < ROW_VALUES) {                 /* :12-26: a value ends at ',' or at a newline that follows at least one character */
		int ch = fgetc(csv->file);
		if (ch == EOF) return 1;                  /* (the reference spins here on a truncated row; a short file is reported instead) */
		if (ch == ',' || (ch == '\n' && have > 0)) {
			token[have] = '\0';
			csv->buffer[filled++] = (float)atof(token);
			have = 0;
		} else if (ch != '\n' && have < (int)sizeof(token) - 1) {
			token[have++] = (char)ch;
		}
	}
	return 0;
}

void visualize_digit_data(struct MnistCSV* csv) {
	const float* px = csv->buffer + 1;            /* buffer[0] is the label */
	printf("============================\n");
	printf("Data for digit %.f:\n", csv->buffer[0]);
	for (int r = 0; r < SIDE; r++) {
		for (int c = 0; c < SIDE; c++) {
			float v = px[r * SIDE + c];
			putchar(v < 0.32 ? ' ' : (v < 0.6 ? ':' : '#'));
		}
		putchar('\n');
	}
	printf("============================\n");
}
