/* cifar_unet_gpu.c -- the reference's CIFAR-10 diffusion U-Net program (model/cifar_unet.c: `init`, `train <epochs>`, `run [<n>]`) as a C host
 * program over the batched device model of the C-ABI (include/bla.h, bla_unet_*).  Host code stays in C (gcc, C99); every pass runs on the GPU.
 *
 *   reference                                          here
 *   main()  :1940-1966  srand(42), three verbs         the same verbs and usage texts; the rand() stream of srand(42) kept in a state array of
 *                                                      this program's own (the GPU runtime draws from rand() too -- see mnist_nn_gpu.c)
 *   init()  :1846-1856  init_parameters + save         the same draws in the same order (:1804-1844) -- every ResNet block draws its 1x1
 *                                                      residual kernels whether forward() uses them or not (:1470), every up stage its
 *                                                      convolution (:1832,1836,1842) -- and the same file set below data/cifar_unet (:1545-1660)
 *   save_parameters :1545-1660                         byte for byte, the as-written quirks included: down_1/resnet_2 and up_N/resnet_1 are
 *                                                      written with fewer input channels than they have (3 instead of 128; the un-doubled
 *                                                      width), and the mid attention's five files land in mid/ beside an empty
 *                                                      mid/self_attention_0.  BLA_UNET_FULL_FILES=1 writes / reads every input channel instead.
 *   train() :1874-1934  ONE example: init_parameters,  `train <passes> [<batch>]`: init_parameters, then per pass `batch` examples, each drawn
 *     load_example, Gaussian noise, forward, MSE,      the way one call of the reference's train() draws its one: fill_random_data (rand()),
 *     backward; no update (the moment buffers are      3 x 32 x 32 values of random_gaussian on rand_r(&seed) with seed 0 carried on, the
 *     allocated and never used)                        dropout decisions of the 18 blocks in forward order (_dropout :1032-1042: one rand()
 *                                                      per element of the block's second ReLU).  One batched forward + backward on the GPU;
 *                                                      loss = mean over the batch of compute_mse_loss (:1858-1872); gradients are the sum over
 *                                                      the images.  `train 1` (batch 1) is exactly the reference's train().  The reference
 *                                                      has no update step; BLA_UNET_LEARN_RATE=x adds plain SGD (params -= x / batch * grads).
 *   run()   :1936-1938  empty                          `run [<n>]`: parameters from the files, n examples through forward(), mean loss printed
 *   time embedding: malloc'd, never written (:535)     zeros; BLA_UNET_TIMESTEP=t: relu(sinusoidal embedding of t) ("passed through ReLU
 *                                                      already", :168)
 *   fan_in = height x width of the resolution          the default.  With it the reference's own backward pass leaves fp32's range: group_norm
 *     (:1455,1474), whatever the channel count         divides by the VARIANCE (lib/norm.c:36-44), activations drift, and the gradient of the
 *                                                      reference's init reaches 1e72 in the reference's own fp64 arithmetic (measured with the
 *                                                      oracle, tests/test_c_unet.py).  BLA_UNET_INIT=unit draws the same rand() values onto
 *                                                      +-sqrt(3 / (values one output sums over)) instead: variance preserving, trainable.
 *   draws <images> <dir>                               host only (tests): everything `train 1 <images>` would hand the device, as raw files
 *
 * BLA_UNET_DUMP=<dir> makes train write what it uploaded (params, x, time embedding, noise, dropout decisions) and what came back (prediction,
 * gradient bucket) as raw little-endian files; tests/test_c_unet.py compares those with the oracle.
 *
 * Paths: data/cifar_unet/... and data/cifar/data_batch_1.bin relative to the working directory, as in the reference (:49,1878);
 * BLA_UNET_WEIGHTS / BLA_CIFAR_BATCH override the directory / the file.
 *
 *   gcc -std=c99 -O2 -I include -I big-linear-algebra_amd/lib examples/cifar_unet_gpu.c -o cifar_unet_gpu \
 *       -L big-linear-algebra_amd/lib -l:libbla_host.so -L big-linear-algebra_amd/csrc -l:libbla_hip.so -lm */
#define _XOPEN_SOURCE 600      /* initstate / setstate, rand_r */
#include "bla.h"
#include "cifar10.h"
#include "csv.h"
#include "util.h"
#include <errno.h>
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

/* model/cifar_unet.c:26-37 */
enum { IMAGE_SIDE = 32, IMAGE_CHANNELS = 3, TIME_EMBED_DIM = 512, KERNEL_SIZE = 3, GROUP_SIZE = 32, KEY_DIM = 16, RESIZE_STRIDE = 2 };
static const int kDims[4] = {128, 256, 256, 256};
static const float DROPOUT_RATE = 0.1;
enum { IMAGE_FLOATS = IMAGE_CHANNELS * IMAGE_SIDE * IMAGE_SIDE, MAX_TENSORS = 160, MAX_BLOCKS = 18 };

#define CHECK(call)                                                                              \
	do {                                                                                         \
		bla_status st_ = (call);                                                                 \
		if (st_ != BLA_OK) {                                                                     \
			fprintf(stderr, "%s failed: %s (%s)\n", #call, bla_status_string(st_), bla_last_error()); \
			exit(1);                                                                             \
		}                                                                                        \
	} while (0)

static char g_rng_mine[128];
static char* g_rng_others;
static void rng_begin(void) { g_rng_others = setstate(g_rng_mine); }
static void rng_end(void) { (void)setstate(g_rng_others); }
static void rng_seed(unsigned seed) { g_rng_others = initstate(seed, g_rng_mine, sizeof g_rng_mine); rng_end(); }

static const char* env_or(const char* name, const char* fallback) { const char* v = getenv(name); return v && *v ? v : fallback; }
static int env_flag(const char* name) { const char* v = getenv(name); return v && *v && strcmp(v, "0") != 0; }
static int side_of(int resolution) { int s = IMAGE_SIDE; for (int r = 1; r < resolution; r++) s = (s + RESIZE_STRIDE - 1) / RESIZE_STRIDE; return s; }   /* :39-46 */

/* ---- the parameter tensors, in init_parameters' order (:1804-1844) ------------------------------------------------------------------ */
typedef enum { DRAW_HE, DRAW_XAVIER, DRAW_ZERO } Draw;
typedef struct Tensor {
	char name[64];       /* the device model's name for it (the reference's struct members), "" = none */
	char file[96];       /* below the data directory */
	int rows, cols;      /* conv kernels: rows = out x in matrices of cols = k x k values ([out][in][k][k]); matrices: as allocated */
	int in, in_written;  /* conv kernels: input channels held / input channels save_parameters writes; 0 for a matrix */
	Draw draw;
	int fan_in, fan_out;
	int true_fan_in;     /* the values one output sums over: in x k x k of a kernel set, the rows of a matrix */
	float* host;         /* rows x cols */
} Tensor;
static Tensor g_tensors[MAX_TENSORS];
static int g_tensor_count;
static struct { int channels, side; } g_blocks[MAX_BLOCKS];   /* ResNet blocks in forward order: the shape of their dropout decisions */
static int g_block_count;
static char g_dirs[48][96];                                   /* directories save_parameters makes, in its order */
static int g_dir_count;

static Tensor* plan(const char* name, const char* file, int rows, int cols, int in, int in_written, Draw draw, int fan_in, int fan_out) {
	if (g_tensor_count == MAX_TENSORS) { fprintf(stderr, "tensor table full\n"); exit(1); }
	Tensor* t = &g_tensors[g_tensor_count++];
	snprintf(t->name, sizeof t->name, "%s", name);
	snprintf(t->file, sizeof t->file, "%s", file);
	t->rows = rows; t->cols = cols; t->in = in; t->in_written = in_written; t->draw = draw; t->fan_in = fan_in; t->fan_out = fan_out;
	t->true_fan_in = in ? in * cols : rows;
	t->host = calloc((size_t)rows * cols, sizeof(float));
	return t;
}
static void plan_dir(const char* dir) { snprintf(g_dirs[g_dir_count++], sizeof g_dirs[0], "%s", dir); }
static void plan_conv(const char* name, const char* file, int side, int in, int out, int k, int in_written) {
	plan(name, file, out * in, k * k, in, in_written, DRAW_HE, side * side, 0);          /* _init_conv_kernels :1454-1461: fan_in = height x width */
}
/* _init_resnet_block :1463-1471 / _save_resnet_block :1511-1526 */
static void plan_resnet(const char* member, const char* dir, int resolution, int in, int out, int in_written) {
	char name[64], file[96];
	const int side = side_of(resolution);
	plan_dir(dir);
	snprintf(name, sizeof name, "%s.conv_1_kernels", member); snprintf(file, sizeof file, "%s/conv_1.csv", dir);
	plan_conv(name, file, side, in, out, KERNEL_SIZE, in_written);
	snprintf(name, sizeof name, "%s.conv_2_kernels", member); snprintf(file, sizeof file, "%s/conv_2.csv", dir);
	plan_conv(name, file, side, out, out, KERNEL_SIZE, out);
	snprintf(name, sizeof name, "%s.time_weights", member); snprintf(file, sizeof file, "%s/time_weight.csv", dir);
	plan(name, file, TIME_EMBED_DIM, out, 0, 0, DRAW_HE, TIME_EMBED_DIM, 0);
	snprintf(name, sizeof name, "%s.time_biases", member); snprintf(file, sizeof file, "%s/time_bias.csv", dir);
	plan(name, file, 1, out, 0, 0, DRAW_ZERO, 0, 0);
	snprintf(name, sizeof name, "%s.residual_conv_kernels", member); snprintf(file, sizeof file, "%s/conv_3.csv", dir);
	plan_conv(in != out ? name : "", file, side, in, out, 1, in_written);                 /* drawn and saved always; used when in != out (:1062) */
	g_blocks[g_block_count].channels = out; g_blocks[g_block_count].side = side; g_block_count++;
}
/* _init_self_attention_block :1473-1482 / _save_self_attention_block :1528-1543; `dir` is made, the files go below `files_in` */
static void plan_attention(const char* member, const char* dir, const char* files_in, int resolution, int embed) {
	static const char* part[5] = {"Q_proj", "K_proj", "V_proj", "weights", "biases"}, *leaf[5] = {"query", "key", "value", "weight", "bias"};
	const int area = side_of(resolution) * side_of(resolution);
	plan_dir(dir);
	for (int i = 0; i < 5; i++) {
		char name[64], file[96];
		snprintf(name, sizeof name, "%s.%s", member, part[i]); snprintf(file, sizeof file, "%s/%s.csv", files_in, leaf[i]);
		if (i < 2) plan(name, file, embed, KEY_DIM, 0, 0, DRAW_XAVIER, area, KEY_DIM);
		else if (i == 2) plan(name, file, embed, KEY_DIM, 0, 0, DRAW_HE, area, 0);
		else if (i == 3) plan(name, file, KEY_DIM, embed, 0, 0, DRAW_HE, KEY_DIM, 0);
		else plan(name, file, 1, embed, 0, 0, DRAW_ZERO, 0, 0);
	}
}
static void plan_model(void) {
	const int* D = kDims;
	const int full = env_flag("BLA_UNET_FULL_FILES");
	plan_dir("down_1");
	plan_resnet("down_1_resnet_1", "down_1/resnet_1", 1, IMAGE_CHANNELS, D[0], IMAGE_CHANNELS);
	plan_resnet("down_1_resnet_2", "down_1/resnet_2", 1, D[0], D[0], full ? D[0] : IMAGE_CHANNELS);        /* :1558 writes 3 input channels */
	plan_conv("down_1_conv_kernels", "down_1/conv_0.csv", side_of(1), D[0], D[1], KERNEL_SIZE, D[0]);
	plan_dir("down_2");
	plan_resnet("down_2_resnet_1", "down_2/resnet_1", 2, D[1], D[1], D[1]);
	plan_attention("down_2_self_attention_1", "down_2/self_attention_1", "down_2/self_attention_1", 2, D[1]);
	plan_resnet("down_2_resnet_2", "down_2/resnet_2", 2, D[1], D[1], D[1]);
	plan_attention("down_2_self_attention_2", "down_2/self_attention_2", "down_2/self_attention_2", 2, D[1]);
	plan_conv("down_2_conv_kernels", "down_2/conv_0.csv", side_of(2), D[1], D[2], KERNEL_SIZE, D[1]);
	plan_dir("down_3");
	plan_resnet("down_3_resnet_1", "down_3/resnet_1", 3, D[2], D[2], D[2]);
	plan_resnet("down_3_resnet_2", "down_3/resnet_2", 3, D[2], D[2], D[2]);
	plan_conv("down_3_conv_kernels", "down_3/conv_0.csv", side_of(3), D[2], D[3], KERNEL_SIZE, D[2]);
	plan_dir("down_4");
	plan_resnet("down_4_resnet_1", "down_4/resnet_1", 4, D[3], D[3], D[3]);
	plan_resnet("down_4_resnet_2", "down_4/resnet_2", 4, D[3], D[3], D[3]);
	plan_dir("mid");
	plan_resnet("mid_resnet_1", "mid/resnet_1", 4, D[3], D[3], D[3]);
	plan_attention("mid_self_attention", "mid/self_attention_0", full ? "mid/self_attention_0" : "mid", 4, D[3]);   /* :1612: position of mid/ */
	plan_resnet("mid_resnet_2", "mid/resnet_2", 4, D[3], D[3], D[3]);
	plan_dir("up_1");
	plan_resnet("up_1_resnet_1", "up_1/resnet_1", 4, 2 * D[3], D[3], full ? 2 * D[3] : D[3]);             /* :1621 and the other up_N/resnet_1: un-doubled */
	plan_resnet("up_1_resnet_2", "up_1/resnet_2", 4, D[3], D[3], D[3]);
	plan_conv(D[3] != D[2] ? "up_1_conv_kernels" : "", "up_1/conv_0.csv", side_of(3), D[3], D[2], KERNEL_SIZE, D[3]);
	plan_dir("up_2");
	plan_resnet("up_2_resnet_1", "up_2/resnet_1", 3, 2 * D[2], D[2], full ? 2 * D[2] : D[2]);
	plan_resnet("up_2_resnet_2", "up_2/resnet_2", 3, D[2], D[2], D[2]);
	plan_conv(D[2] != D[1] ? "up_2_conv_kernels" : "", "up_2/conv_0.csv", side_of(2), D[2], D[1], KERNEL_SIZE, D[2]);
	plan_dir("up_3");
	plan_resnet("up_3_resnet_1", "up_3/resnet_1", 2, 2 * D[1], D[1], full ? 2 * D[1] : D[1]);
	plan_attention("up_3_self_attention_1", "up_3/self_attention_1", "up_3/self_attention_1", 2, D[1]);
	plan_resnet("up_3_resnet_2", "up_3/resnet_2", 2, D[1], D[1], D[1]);
	plan_attention("up_3_self_attention_2", "up_3/self_attention_2", "up_3/self_attention_2", 2, D[1]);
	plan_conv(D[1] != D[0] ? "up_3_conv_kernels" : "", "up_3/conv_0.csv", side_of(1), D[1], D[0], KERNEL_SIZE, D[1]);
	plan_dir("up_4");
	plan_resnet("up_4_resnet_1", "up_4/resnet_1", 1, 2 * D[0], D[0], full ? 2 * D[0] : D[0]);
	plan_resnet("up_4_resnet_2", "up_4/resnet_2", 1, D[0], D[0], D[0]);
	plan_conv("output_conv_kernels", "output_conv.csv", side_of(1), D[0], IMAGE_CHANNELS, KERNEL_SIZE, D[0]);
}

/* init_parameters :1804-1844: _init_params_he / _init_params_xavier (:1439-1452) in double, stored as the float save_parameters would write */
static void draw_parameters(void) {
	const int unit_gain = strcmp(env_or("BLA_UNET_INIT", "reference"), "unit") == 0;
	rng_begin();
	for (int t = 0; t < g_tensor_count; t++) {
		Tensor* x = &g_tensors[t];
		const size_t n = (size_t)x->rows * x->cols;
		if (x->draw == DRAW_ZERO) { memset(x->host, 0, n * sizeof(float)); continue; }
		double scale = x->draw == DRAW_HE ? sqrt(6.0 / x->fan_in) : sqrt(6.0 / (x->fan_in + x->fan_out));
		if (unit_gain) scale = sqrt(3.0 / x->true_fan_in);
		for (size_t i = 0; i < n; i++) x->host[i] = (float)(2 * scale * (double)rand() / RAND_MAX - scale);
	}
	rng_end();
}

static void data_path(char* out, size_t n, const char* below) { snprintf(out, n, "%s%s%s", env_or("BLA_UNET_WEIGHTS", "data/cifar_unet"), *below ? "/" : "", below); }

/* save_parameters :1545-1660 */
static void save_parameters(void) {
	char path[512];
	data_path(path, sizeof path, ""); mkdir(path, 0777);
	for (int d = 0; d < g_dir_count; d++) { data_path(path, sizeof path, g_dirs[d]); mkdir(path, 0777); }
	for (int t = 0; t < g_tensor_count; t++) {
		const Tensor* x = &g_tensors[t];
		data_path(path, sizeof path, x->file);
		if (!x->in || x->in_written == x->in) { write_csv_contents(path, x->host, x->cols, x->rows); continue; }
		/* _save_conv_kernels with fewer input channels than the kernels have (:1493-1509): the first in_written of every output channel */
		const int out = x->rows / x->in;
		float* part = malloc((size_t)out * x->in_written * x->cols * sizeof(float));
		for (int o = 0; o < out; o++) memcpy(part + (size_t)o * x->in_written * x->cols, x->host + (size_t)o * x->in * x->cols, (size_t)x->in_written * x->cols * sizeof(float));
		write_csv_contents(path, part, x->cols, out * x->in_written);
		free(part);
	}
}
/* load_parameters :1720-1802 (the same channel counts as save_parameters); input channels a file does not hold stay zero */
static void load_parameters(void) {
	char path[512];
	for (int t = 0; t < g_tensor_count; t++) {
		Tensor* x = &g_tensors[t];
		data_path(path, sizeof path, x->file);
		FILE* f = fopen(path, "r");
		if (!f) { fprintf(stderr, "cannot open %s (run `init` first)\n", path); exit(1); }
		int count = 0;
		float* v = read_csv_contents_file(f, &count);
		const int in_file = x->in ? x->in_written : 0, out = x->in ? x->rows / x->in : 0;
		const size_t want = x->in ? (size_t)out * in_file * x->cols : (size_t)x->rows * x->cols;
		if ((size_t)count != want) { fprintf(stderr, "%s holds %d values, expected %zu\n", path, count, want); exit(1); }
		memset(x->host, 0, (size_t)x->rows * x->cols * sizeof(float));
		if (!x->in) memcpy(x->host, v, want * sizeof(float));
		else for (int o = 0; o < out; o++) memcpy(x->host + (size_t)o * x->in * x->cols, v + (size_t)o * in_file * x->cols, (size_t)in_file * x->cols * sizeof(float));
		free(v);
	}
}

/* ---- one example the way train() makes it (:1903-1914) + the dropout decisions forward() would draw (:1032-1042) ---------------------- */
typedef struct Inputs {
	int batch;
	size_t drop_per_image;       /* sum over the blocks */
	float *x, *noise, *temb;     /* [B][3][32][32], [B][3][32][32], [B][512] */
	unsigned char* drop;         /* device layout: block by block, inside a block image by image */
} Inputs;
static Inputs inputs_alloc(int batch) {
	Inputs in; in.batch = batch; in.drop_per_image = 0;
	for (int b = 0; b < g_block_count; b++) in.drop_per_image += (size_t)g_blocks[b].channels * g_blocks[b].side * g_blocks[b].side;
	in.x = malloc((size_t)batch * IMAGE_FLOATS * sizeof(float)); in.noise = malloc((size_t)batch * IMAGE_FLOATS * sizeof(float));
	in.temb = calloc((size_t)batch * TIME_EMBED_DIM, sizeof(float)); in.drop = malloc(in.drop_per_image * batch);
	return in;
}
static void inputs_free(Inputs* in) { free(in->x); free(in->noise); free(in->temb); free(in->drop); }
static void time_embedding(float* out) {
	const char* v = getenv("BLA_UNET_TIMESTEP");
	memset(out, 0, TIME_EMBED_DIM * sizeof(float));
	if (!v || !*v) return;
	const double t = atof(v); const int half = TIME_EMBED_DIM / 2;
	for (int i = 0; i < half; i++) {
		const double w = exp(-log(10000.0) * i / half), s = sin(t * w), c = cos(t * w);
		out[i] = (float)(s > 0 ? s : 0); out[half + i] = (float)(c > 0 ? c : 0);
	}
}
static void draw_inputs(Inputs* in, int fd, unsigned int* noise_seed, int with_dropout) {
	uint8_t pixels[3072];
	rng_begin();
	for (int b = 0; b < in->batch; b++) {
		fill_random_data(fd, pixels);                                                           /* load_example :221-233 */
		for (int i = 0; i < IMAGE_FLOATS; i++) in->x[(size_t)b * IMAGE_FLOATS + i] = (float)(((double)pixels[i] - 127.5) / 127.5);
		for (int i = 0; i < IMAGE_FLOATS; i++) in->noise[(size_t)b * IMAGE_FLOATS + i] = (float)random_gaussian(noise_seed);
		time_embedding(in->temb + (size_t)b * TIME_EMBED_DIM);
		size_t at = 0;
		for (int k = 0; k < g_block_count; k++) {
			const size_t n = (size_t)g_blocks[k].channels * g_blocks[k].side * g_blocks[k].side;
			unsigned char* d = in->drop + at * in->batch + (size_t)b * n;
			for (size_t i = 0; i < n; i++) d[i] = with_dropout ? (float)rand() / RAND_MAX < DROPOUT_RATE : 0;
			at += n;
		}
	}
	rng_end();
}
static int open_batch_file(void) {
	const char* path = env_or("BLA_CIFAR_BATCH", "data/cifar/data_batch_1.bin");                /* :1878 */
	const int fd = open(path, O_RDONLY);
	if (fd < 0) { fprintf(stderr, "cannot open %s: %s\n", path, strerror(errno)); exit(1); }
	return fd;
}
static void dump(const char* dir, const char* leaf, const void* data, size_t bytes) {
	char path[512];
	snprintf(path, sizeof path, "%s/%s", dir, leaf);
	FILE* f = fopen(path, "wb");
	if (!f || fwrite(data, 1, bytes, f) != bytes) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
	fclose(f);
}

/* ---- the device model ----------------------------------------------------------------------------------------------------------------- */
typedef struct Device {
	bla_unet* net;
	int batch;
	float *x, *noise, *temb;
	unsigned char* drop;
	float* bucket;     /* host image of the parameter bucket */
} Device;
static Device device_open(int batch, size_t drop_per_image) {
	Device dv; memset(&dv, 0, sizeof dv); dv.batch = batch;
	CHECK(bla_init(atoi(env_or("BLA_DEVICE", "0"))));
	bla_unet_config cfg = {IMAGE_SIDE, IMAGE_SIDE, IMAGE_CHANNELS, {kDims[0], kDims[1], kDims[2], kDims[3]}, TIME_EMBED_DIM, KERNEL_SIZE, GROUP_SIZE, KEY_DIM};
	CHECK(bla_unet_create_batched(&dv.net, &cfg, batch));
	if (bla_unet_dropout_count(dv.net) != drop_per_image * batch) { fprintf(stderr, "dropout layout: device %zu, host %zu\n", bla_unet_dropout_count(dv.net), drop_per_image * batch); exit(1); }
	CHECK(bla_malloc((void**)&dv.x, (size_t)batch * IMAGE_FLOATS * sizeof(float)));
	CHECK(bla_malloc((void**)&dv.noise, (size_t)batch * IMAGE_FLOATS * sizeof(float)));
	CHECK(bla_malloc((void**)&dv.temb, (size_t)batch * TIME_EMBED_DIM * sizeof(float)));
	CHECK(bla_malloc((void**)&dv.drop, drop_per_image * batch));
	dv.bucket = calloc(bla_unet_param_count(dv.net), sizeof(float));
	return dv;
}
/* host tensors -> bucket -> device; every device tensor must be fed by exactly one host tensor */
static void device_set_params(Device* dv) {
	const int n = bla_unet_tensor_count(dv->net);
	int fed = 0;
	for (int i = 0; i < n; i++) {
		size_t off, count; char name[64];
		CHECK(bla_unet_tensor_info(dv->net, i, &off, &count, name, sizeof name));
		for (int t = 0; t < g_tensor_count; t++) {
			const Tensor* x = &g_tensors[t];
			if (strcmp(x->name, name) != 0) continue;
			if ((size_t)x->rows * x->cols != count) { fprintf(stderr, "%s: device %zu values, host %d\n", name, count, x->rows * x->cols); exit(1); }
			memcpy(dv->bucket + off, x->host, count * sizeof(float)); fed++;
		}
	}
	if (fed != n) { fprintf(stderr, "%d of the device model's %d tensors have a host tensor\n", fed, n); exit(1); }
	CHECK(bla_memcpy_h2d(bla_unet_params(dv->net), dv->bucket, bla_unet_param_count(dv->net) * sizeof(float), NULL));
}
static void device_upload(Device* dv, const Inputs* in) {
	CHECK(bla_memcpy_h2d(dv->x, in->x, (size_t)in->batch * IMAGE_FLOATS * sizeof(float), NULL));
	CHECK(bla_memcpy_h2d(dv->noise, in->noise, (size_t)in->batch * IMAGE_FLOATS * sizeof(float), NULL));
	CHECK(bla_memcpy_h2d(dv->temb, in->temb, (size_t)in->batch * TIME_EMBED_DIM * sizeof(float), NULL));
	CHECK(bla_memcpy_h2d(dv->drop, in->drop, in->drop_per_image * in->batch, NULL));
}
static void device_close(Device* dv) {
	CHECK(bla_free(dv->x)); CHECK(bla_free(dv->noise)); CHECK(bla_free(dv->temb)); CHECK(bla_free(dv->drop));
	CHECK(bla_unet_destroy(dv->net)); free(dv->bucket);
	CHECK(bla_shutdown());
}
/* compute_mse_loss :1858-1872 per image (float accumulation in the reference's order), averaged over the batch */
static float batch_loss(const float* prediction, const float* noise, int batch) {
	double total = 0;
	for (int b = 0; b < batch; b++) {
		float loss = 0;
		for (int i = 0; i < IMAGE_FLOATS; i++) { const float r = prediction[(size_t)b * IMAGE_FLOATS + i] - noise[(size_t)b * IMAGE_FLOATS + i]; loss += r * r; }
		total += loss / IMAGE_FLOATS;
	}
	return (float)(total / batch);
}

static void init(void) {
	draw_parameters();
	save_parameters();
}

static void train(int passes, int batch) {
	const int fd = open_batch_file();
	const char* dump_dir = getenv("BLA_UNET_DUMP");
	const double learn_rate = atof(env_or("BLA_UNET_LEARN_RATE", "0"));
	draw_parameters();                                                                          /* :1900 (load_parameters is commented out there) */
	Inputs in = inputs_alloc(batch);
	Device dv = device_open(batch, in.drop_per_image);
	device_set_params(&dv);
	const size_t params = bla_unet_param_count(dv.net);
	float* prediction = malloc((size_t)batch * IMAGE_FLOATS * sizeof(float));
	unsigned int seed = 0;                                                                      /* :1902 */
	for (int pass = 0; pass < passes; pass++) {
		draw_inputs(&in, fd, &seed, 1);
		device_upload(&dv, &in);
		CHECK(bla_unet_forward_f32(dv.net, NULL, dv.x, dv.temb, dv.drop));
		CHECK(bla_unet_backward_f32(dv.net, NULL, dv.noise));
		CHECK(bla_memcpy_d2h(prediction, bla_unet_output(dv.net), (size_t)batch * IMAGE_FLOATS * sizeof(float), NULL));
		CHECK(bla_stream_sync(NULL));
		printf("Pass %d:\tAvg loss: %f\n", pass, batch_loss(prediction, in.noise, batch));
		if (dump_dir && pass == passes - 1) {
			float* grads = malloc(params * sizeof(float));
			CHECK(bla_memcpy_d2h(grads, bla_unet_grads(dv.net), params * sizeof(float), NULL));
			CHECK(bla_stream_sync(NULL));
			CHECK(bla_memcpy_d2h(dv.bucket, bla_unet_params(dv.net), params * sizeof(float), NULL));
			CHECK(bla_stream_sync(NULL));
			dump(dump_dir, "params.f32", dv.bucket, params * sizeof(float)); dump(dump_dir, "grads.f32", grads, params * sizeof(float));
			dump(dump_dir, "x.f32", in.x, (size_t)batch * IMAGE_FLOATS * sizeof(float)); dump(dump_dir, "noise.f32", in.noise, (size_t)batch * IMAGE_FLOATS * sizeof(float));
			dump(dump_dir, "temb.f32", in.temb, (size_t)batch * TIME_EMBED_DIM * sizeof(float)); dump(dump_dir, "drop.u8", in.drop, in.drop_per_image * batch);
			dump(dump_dir, "prediction.f32", prediction, (size_t)batch * IMAGE_FLOATS * sizeof(float));
			free(grads);
		}
		if (learn_rate != 0) {                                                                  /* not in the reference: plain SGD on the batch mean */
			CHECK(bla_scale_f32(NULL, bla_unet_grads(dv.net), params, (float)(-learn_rate / batch)));
			CHECK(bla_add_f32(NULL, bla_unet_params(dv.net), bla_unet_grads(dv.net), params));
		}
	}
	if (learn_rate != 0) {                                                                      /* trained parameters back into the file set */
		CHECK(bla_memcpy_d2h(dv.bucket, bla_unet_params(dv.net), params * sizeof(float), NULL));
		CHECK(bla_stream_sync(NULL));
		const int n = bla_unet_tensor_count(dv.net);
		for (int i = 0; i < n; i++) {
			size_t off, count; char name[64];
			CHECK(bla_unet_tensor_info(dv.net, i, &off, &count, name, sizeof name));
			for (int t = 0; t < g_tensor_count; t++) if (strcmp(g_tensors[t].name, name) == 0) memcpy(g_tensors[t].host, dv.bucket + off, count * sizeof(float));
		}
		save_parameters();
	}
	free(prediction); inputs_free(&in); device_close(&dv); close(fd);
}

static void run(int num_predictions) {
	const int fd = open_batch_file();
	int batch = atoi(env_or("BLA_UNET_BATCH", "16"));
	if (batch > num_predictions) batch = num_predictions;
	if (batch < 1) { close(fd); return; }
	load_parameters();
	Inputs in = inputs_alloc(batch);
	Device dv = device_open(batch, in.drop_per_image);
	device_set_params(&dv);
	float* prediction = malloc((size_t)batch * IMAGE_FLOATS * sizeof(float));
	unsigned int seed = 0;
	double total = 0; int done = 0;
	printf("Predicting the noise of %d examples...", num_predictions);
	while (done < num_predictions) {
		draw_inputs(&in, fd, &seed, 0);                                                         /* inference: nothing dropped */
		device_upload(&dv, &in);
		CHECK(bla_unet_forward_f32(dv.net, NULL, dv.x, dv.temb, NULL));
		CHECK(bla_memcpy_d2h(prediction, bla_unet_output(dv.net), (size_t)batch * IMAGE_FLOATS * sizeof(float), NULL));
		CHECK(bla_stream_sync(NULL));
		const int take = num_predictions - done < batch ? num_predictions - done : batch;       /* a last, partial batch counts its first images */
		total += (double)batch_loss(prediction, in.noise, take) * take; done += take;
	}
	printf("done! Avg loss: %f\n", total / num_predictions);
	free(prediction); inputs_free(&in); device_close(&dv); close(fd);
}

/* host only: what `train 1 <images>` would upload, and the rand() value that would come next */
static void draws(int images, const char* dir) {
	const int fd = open_batch_file();
	draw_parameters();
	Inputs in = inputs_alloc(images);
	unsigned int seed = 0;
	draw_inputs(&in, fd, &seed, 1);
	dump(dir, "x.f32", in.x, (size_t)images * IMAGE_FLOATS * sizeof(float)); dump(dir, "noise.f32", in.noise, (size_t)images * IMAGE_FLOATS * sizeof(float));
	dump(dir, "drop.u8", in.drop, in.drop_per_image * images);
	char path[512];
	snprintf(path, sizeof path, "%s/params.f32", dir);                                          /* the tensors the device model has, in init order */
	FILE* f = fopen(path, "wb");
	for (int t = 0; f && t < g_tensor_count; t++)
		if (g_tensors[t].name[0]) fwrite(g_tensors[t].host, sizeof(float), (size_t)g_tensors[t].rows * g_tensors[t].cols, f);
	if (!f || fclose(f) != 0) { fprintf(stderr, "cannot write %s\n", path); exit(1); }
	rng_begin(); const int next = rand(); rng_end();
	printf("blocks %d drop_per_image %zu next_rand %d\n", g_block_count, in.drop_per_image, next);
	inputs_free(&in); close(fd);
}

int main(int argc, char** argv) {
	rng_seed(42);                                                                               /* srand(42), :1941 */
	if (argc < 2) {
		printf("Please supply an argument, options:\n\trun [<num samples> (default 1)]\n\ttrain <num epochs>\n\tinit\n");
		exit(1);
	}
	plan_model();
	if (strncmp(argv[1], "run", 3) == 0) {
		run(argc < 3 ? 1 : atoi(argv[2]));
	} else if (strncmp(argv[1], "train", 5) == 0) {
		if (argc < 3) {
			printf("Please supply a number of epochs, usage:\n\ttrain <num_epochs>\n");
			exit(1);
		}
		train(atoi(argv[2]), argc < 4 ? 1 : atoi(argv[3]));
	} else if (strncmp(argv[1], "init", 4) == 0) {
		init();
	} else if (strcmp(argv[1], "draws") == 0 && argc >= 4) {
		draws(atoi(argv[2]), argv[3]);
	} else {
		printf("Unrecognized argument, options:\n\trun [<num samples> (default 1)]\n\ttrain <num epochs>\n\tinit\n");
		exit(1);
	}
	return 0;
}
