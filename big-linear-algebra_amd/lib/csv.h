/* csv.h -- drop-in for the reference's lib/csv.h: the text I/O the models use for weights and datasets (host only).
 *
 * File format, as the reference writes and reads it (lib/csv.c:18-69): every value is followed by a comma, including
 * the last one of a row ("%f," per value, '\n' per row); a reader counts commas to size its buffer.  Returned buffers are
 * malloc'd and owned by the caller. */
#ifndef BLA_DROPIN_CSV_H
#define BLA_DROPIN_CSV_H

#include <stdio.h>

/* number of '\n' in the stream; rewinds it afterwards */
int count_num_lines(FILE* stream);

/* all values of an open stream, in file order; *value_count receives how many.  Closes the stream (as the reference does). */
float* read_csv_contents_file(FILE* stream, int* value_count);

/* convenience: open `path`, read every value, close */
float* read_csv_contents(const char* path);

/* `rows` lines of `cols` values each, "%f," per value -- note the argument order (cols before rows) */
void write_csv_contents(const char* path, float* values, int cols, int rows);

#endif /* BLA_DROPIN_CSV_H */
