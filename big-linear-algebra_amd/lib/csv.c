/* csv.c -- the reference's CSV conventions (lib/csv.c), restated:
 *   - the number of values in a file IS its number of commas (lib/csv.c:7-16), so every value needs a trailing
 *     comma -- which is what write_csv_contents emits ("%f," per value, newline after each row, lib/csv.c:59-69);
 *   - values are parsed with atof into float; '\n' and '\r' never belong to a value; a newline ends a value only
 *     when characters are pending (lib/csv.c:44-53);
 *   - read_csv_contents_file closes the stream it is given (lib/csv.c:55). */
#include "csv.h"
#include <stdlib.h>
#include <string.h>

static int comma_count(FILE* f) {
	int n = 0, ch;
	rewind(f);
	while ((ch = fgetc(f)) != EOF) n += (ch == ',');
	return n;
}

float* read_csv_contents_file(FILE* f, int* num_values) {
	int total = comma_count(f);
	if (num_values) *num_values = total;
	float* values = malloc((total > 0 ? total : 1) * sizeof(float));
	char token[512];
	int len = 0, count = 0, ch;
	rewind(f);
	while ((ch = fgetc(f)) != EOF) {
		if (ch == ',' || (ch == '\n' && len != 0)) {
			token[len] = '\0';
			if (count < total) values[count] = (float)atof(token);
			count++;
			len = 0;
		} else if (ch != '\n' && ch != '\r' && len < (int)sizeof(token) - 1) {
			token[len++] = (char)ch;
		}
	}
	fclose(f);
	return values;
}

float* read_csv_contents(const char* filepath) {
	FILE* f = fopen(filepath, "r");
	if (!f) {   /* the reference dereferences NULL here (lib/csv.c:19-20); fail with a message instead */
		fprintf(stderr, "cannot open CSV file %s\n", filepath);
		exit(1);
	}
	return read_csv_contents_file(f, NULL);
}

void write_csv_contents(const char* filepath, float* data, int cols, int rows) {
	FILE* f = fopen(filepath, "w");
	if (!f) {
		fprintf(stderr, "cannot write CSV file %s\n", filepath);
		exit(1);
	}
	for (int i = 0; i < cols * rows; i++) {
		fprintf(f, "%f,", data[i]);
		if ((i + 1) % cols == 0) fputc('\n', f);
	}
	fclose(f);
}

int count_num_lines(FILE* f) {   /* reference lib/csv.c:72-91: newline count from the current position, -1 on error */
	char buf[1 << 16];
	int lines = 0;
	for (;;) {
		size_t got = fread(buf, 1, sizeof buf, f);
		if (ferror(f)) return -1;
		for (size_t i = 0; i < got; i++) lines += (buf[i] == '\n');
		if (feof(f)) break;
	}
	return lines;
}
