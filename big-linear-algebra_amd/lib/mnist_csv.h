#ifndef __mnist_csv_h__
#define __mnist_csv_h__
/* Drop-in for the reference's legacy streaming reader lib/mnist_csv.h (used by model/mnist_hinge.c and model/mnist.c).  Same include
 * guard and type name as lib/mnist_csv2.h -- the two are mutually exclusive by the reference's design, so this unit is
 * NOT part of libbla_host.so; a program that includes this header compiles lib/mnist_csv.c into itself, as the reference's build does.
 * Host-side dataset plumbing: nothing here touches the device. */
#include <stdio.h>

/* buffer: room for 785 floats (label, then 784 pixels) */
typedef struct MnistCSV {
	FILE* file;
	float* buffer;
	int num_lines;
} MnistCSV;

/* reads the next row (785 comma- or newline-terminated values) into csv->buffer; returns 1 (after printing "CSV file is empty") when
 * the stream is already at end-of-file, else 0 (lib/mnist_csv.c:6-29) */
int get_next_data(struct MnistCSV* csv);
/* 28 x 28 ASCII rendering of buffer[1..784] (' ' < 0.32 <= ':' < 0.6 <= '#') between two rules (lib/mnist_csv.c:31-47) */
void visualize_digit_data(struct MnistCSV* csv);

#endif
