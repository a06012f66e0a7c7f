/* cifar10.h -- drop-in for the reference's lib/cifar10.h: random-access reader of the CIFAR-10 binary batch files (host only).
 *
 * A batch file is 10,000 records of 3,073 bytes: one label byte, then 1,024 red, 1,024 green and 1,024 blue bytes of a
 * 32 x 32 image.  model/cifar_unet.c (load_example, :221-233) opens the file itself and asks for one random image at a
 * time; the constants below are part of that interface. */
#ifndef BLA_DROPIN_CIFAR10_H
#define BLA_DROPIN_CIFAR10_H

#include <stdint.h>
#include <stdio.h>

extern const unsigned int CIFAR10_EXAMPLE_DIM;             /* 32: image side */
extern const unsigned int CIFAR10_NUM_PIXELS;              /* 1,024 per colour plane */
extern const unsigned int CIFAR10_DATA_LENGTH;             /* 3,072 image bytes per record */
extern const unsigned int CIFAR10_LINE_LENGTH;             /* 3,073 bytes per record, label included */
extern const unsigned int CIFAR10_NUM_EXAMPLES_PER_FILE;   /* 10,000 */
extern const unsigned int CIFAR10_BATCH_FILE_SIZE;         /* 30,730,000 */

/* One uniformly drawn record (libc rand(), as in lib/cifar10.c:13-32) -> `pixels`[3072]: red plane, green plane, blue
 * plane, each with its rows flipped top-to-bottom so that a BMP dump shows the image upright.  `fd` is an open batch file. */
void fill_random_data(int fd, uint8_t* pixels);

#endif /* BLA_DROPIN_CIFAR10_H */
