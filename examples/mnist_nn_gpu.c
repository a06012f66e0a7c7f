/* mnist_nn_gpu.c -- the reference's MNIST program (model/mnist_nn.c: `init`, `train <epochs>`, `run [<n>]`) as a C host program over
 * the device-resident trainer of the C-ABI (include/bla.h, bla_mnist_nn_*).  Host code stays in C (gcc, C99); everything per batch
 * runs on the GPU.
 *
 *   reference                                   here
 *   init()   :97-142  He-uniform from rand()    the same draws, the same CSV files (lib/csv.c semantics)
 *   train()  :164-394
 *     load 6 CSVs :165-170                      read_csv_contents -> one flat bucket W1,b1,W2,b2,W3,b3 -> bla_mnist_nn_set_params
 *     mnist_csv_init :180                       the same store (lib/mnist_csv2), uploaded ONCE: the dataset (188 MB for 60,000 rows) stays
 *                                               in HBM in the store's feature-major layout
 *     per epoch: sampler reset :189-191         the same draws from rand() (get_random_data_take's picks, found in O(log N) each),
 *                                               one upload of the epoch's index order
 *     per batch: build input / one-hot :204-217 bla_mnist_nn_gather_batch (device)
 *                scale, forward, metrics,       bla_mnist_nn_fused_step: six launches; loss / accuracy accumulate on the device inside the
 *                backward, update :218-315      output layer's launch (bla_mnist_nn_metrics_*), read back once per epoch
 *     epoch line :341                           the same printf
 *     save 6 CSVs :345-376                      the same write_csv_contents calls
 *   run()    :401-510                           forward passes in chunks + the same counting; the same two printf texts
 *
 * Differences, stated: (1) the batch size is a run-time argument (the reference compiles in SGD_BATCH_SIZE 64); (2) bias gradients are
 * true row sums -- matrix_col_sum as written reads out of bounds at the reference's own batch size; `as-written` selects the
 * literal form where it is defined (every layer width <= batch); (3) fp32 on the device (the reference computes in double, :4 of matrix.h).
 *
 * Data parallel (BASELINE configs[3]): BLA_GPUS=R drives R GPUs from this one process -- one bla context and one trainer per GPU, batch
 * columns split R ways, gradients summed by the peer-read exchange kernel (bla_dp_*), identical update everywhere.
 *
 * Paths: data/mnist_nn/{weights,biases}_{1,2,3}.csv and data/mnist/mnist_{train,test}.csv relative to the working directory, as in the
 * reference (:30-35,174,411); BLA_MNIST_WEIGHTS / BLA_MNIST_TRAIN / BLA_MNIST_TEST override the directory / files.
 *
 *   gcc -std=c99 -O2 -I include -I big-linear-algebra_amd/lib examples/mnist_nn_gpu.c -o mnist_nn_gpu \
 *       -L big-linear-algebra_amd/lib -l:libbla_host.so -L big-linear-algebra_amd/csrc -l:libbla_hip.so -lm */
#define _XOPEN_SOURCE 600      /* clock_gettime, initstate / setstate */
#include "bla.h"
#include "csv.h"
#include "mnist_csv2.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#define SGD_LEARN_RATE_MULTIPLIER 0.02      /* model/mnist_nn.c:12 */
enum { N0 = 784, N1 = 256, N2 = 128, N3 = 10, MAX_GPUS = 16 };
static const int kSizes[4] = {N0, N1, N2, N3};
static const int kRows[6] = {N1, N1, N2, N2, N3, N3}, kCols[6] = {N0, 1, N1, 1, N2, 1};   /* W1,b1,W2,b2,W3,b3 */
static const char* kFiles[6] = {"weights_1.csv", "biases_1.csv", "weights_2.csv", "biases_2.csv", "weights_3.csv", "biases_3.csv"};

#define CHECK(call)                                                                              \
	do {                                                                                         \
		bla_status st_ = (call);                                                                 \
		if (st_ != BLA_OK) {                                                                     \
			fprintf(stderr, "%s failed: %s (%s)\n", #call, bla_status_string(st_), bla_last_error()); \
			exit(1);                                                                             \
		}                                                                                        \
	} while (0)

/* The reference's draws come from libc rand() after srand(42) (model/mnist_nn.c:513).  The GPU runtime and the collective library are
 * free to call rand() / srand() themselves (observed: the example order changed from run to run once the device was initialised), so this
 * program keeps the stream the REFERENCE would see in a state array of its own -- glibc's rand() is random() on the current state, and
 * initstate(42, 128-byte array) is exactly srand(42) -- and switches to it only around its own draws. */
static char g_rng_mine[128];
static char* g_rng_others;
static void rng_begin(void) { g_rng_others = setstate(g_rng_mine); }
static void rng_end(void) { (void)setstate(g_rng_others); }
static void rng_seed(unsigned seed) { g_rng_others = initstate(seed, g_rng_mine, sizeof g_rng_mine); rng_end(); }

static const char* env_or(const char* name, const char* fallback) { const char* v = getenv(name); return v && *v ? v : fallback; }
static void weight_path(char* out, size_t n, int which) { snprintf(out, n, "%s/%s", env_or("BLA_MNIST_WEIGHTS", "data/mnist_nn"), kFiles[which]); }
static double now_s(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

/* ---- init: model/mnist_nn.c:97-142 (same order of rand() draws, same float expressions) ---------------------------------------- */
static void init(void) {
	char path[512];
	rng_begin();
	const int fan_in[3] = {N0, N1, N2}, fan_out[3] = {N1, N2, N3};
	for (int l = 0; l < 3; l++) {
		const int count = fan_in[l] * fan_out[l];
		float* w = malloc((size_t)count * sizeof(float));
		float range = 2 * sqrtf(6.0 / (float)fan_in[l]);
		for (int i = 0; i < count; i++) w[i] = range * (float)rand() / (float)(RAND_MAX) - range / 2;
		weight_path(path, sizeof path, 2 * l);
		write_csv_contents(path, w, fan_in[l], fan_out[l]);
		free(w);
	}
	for (int l = 0; l < 3; l++) {
		float* b = calloc((size_t)fan_out[l], sizeof(float));
		weight_path(path, sizeof path, 2 * l + 1);
		write_csv_contents(path, b, 1, fan_out[l]);
		free(b);
	}
	rng_end();
}

static size_t param_count(void) {
	size_t n = 0;
	for (int i = 0; i < 6; i++) n += (size_t)kRows[i] * kCols[i];
	return n;
}

/* the six CSV files -> one flat host bucket in the trainer's order (load_matrix_from_csv x 6, :165-170) */
static float* load_params(void) {
	float* flat = malloc(param_count() * sizeof(float));
	size_t at = 0;
	char path[512];
	for (int i = 0; i < 6; i++) {
		weight_path(path, sizeof path, i);
		float* v = read_csv_contents(path);
		memcpy(flat + at, v, (size_t)kRows[i] * kCols[i] * sizeof(float));
		at += (size_t)kRows[i] * kCols[i];
		free(v);
	}
	return flat;
}

static void save_params(const float* flat) {      /* :371-376: write_csv_contents(file, data, cols, rows) */
	size_t at = 0;
	char path[512];
	for (int i = 0; i < 6; i++) {
		weight_path(path, sizeof path, i);
		write_csv_contents(path, (float*)(flat + at), kCols[i], kRows[i]);
		at += (size_t)kRows[i] * kCols[i];
	}
}

/* ---- the sampler's picks without its O(N) walk ------------------------------------------------------------------------------------
 * get_random_data_take (lib/mnist_csv2.c:41-62) draws n = floor((N - num_sampled) * rand() / RAND_MAX) in float, walks from index 0 until
 * it has passed n not-yet-sampled entries and takes the entry it then stands on (n == 0: entry 0; the entry may already be taken -- the
 * reference's behaviour, kept).  take_order() returns exactly those indices, consuming rand() identically, with a Fenwick tree over the
 * not-yet-sampled flags: "index of the n-th set flag" in O(log N) (tests/test_c_trainer.py compares it with the library's walk). */
static int* take_order(MnistCSV* store, int draws) {
	const int N = store->num_examples;
	int* tree = calloc((size_t)N + 1, sizeof(int));
	int* order = malloc((size_t)(draws > 0 ? draws : 1) * sizeof(int));
	int top = 1;
	while (top * 2 <= N) top *= 2;
	for (int d = 0; d < draws; d++) {
		if (store->num_sampled == store->num_examples || d == 0) {     /* everything taken (or first use): rebuild from the flags */
			if (store->num_sampled == store->num_examples) { memset(store->sampled, 0, (size_t)N); store->num_sampled = 0; }
			memset(tree, 0, ((size_t)N + 1) * sizeof(int));
			for (int i = 1; i <= N; i++) {
				tree[i] += store->sampled[i - 1] ? 0 : 1;
				int up = i + (i & -i);
				if (up <= N) tree[up] += tree[i];
			}
		}
		int n = (int)floor((float)(store->num_examples - store->num_sampled) * (float)rand() / (float)RAND_MAX);
		int at = 0;
		if (n > 0) {       /* smallest prefix holding n set flags: its last element is the n-th not-yet-sampled entry; the walk stands one past it */
			int pos = 0, left = n;
			for (int step = top; step > 0; step >>= 1)
				if (pos + step <= N && tree[pos + step] < left) { pos += step; left -= tree[pos]; }
			at = pos + 1;  /* `pos` entries precede the n-th flag, so it sits at index pos and the walk ends one past it */
			if (pos >= N) at = N;   /* fewer than n flags left: the walk runs off the end */
		}
		if (at >= N) at = N - 1;    /* (the reference writes one past its array here; only reachable when rand() rounds to RAND_MAX in float) */
		if (!store->sampled[at]) {
			store->sampled[at] = 1;
			for (int i = at + 1; i <= N; i += i & -i) tree[i]--;
		}
		store->num_sampled++;
		order[d] = at;
	}
	free(tree);
	return order;
}

/* ---- one replica per GPU ----------------------------------------------------------------------------------------------------------- */
typedef struct Replica {
	bla_context* ctx;       /* NULL: the default context (single GPU) */
	bla_mnist_nn *nn, *tail;     /* full batches / the last, shorter batch of an epoch (shares the buckets) */
	bla_dp* dp;
	float *d_X, *d_y;       /* the dataset, resident */
	int* d_order;           /* the epoch's example order */
} Replica;

static void use(Replica* r) { if (r->ctx) CHECK(bla_context_set_current(r->ctx)); }

static void upload_dataset(Replica* r, const MnistCSV* store) {
	const size_t n = (size_t)store->num_examples;
	CHECK(bla_malloc((void**)&r->d_X, n * N0 * sizeof(float)));
	CHECK(bla_malloc((void**)&r->d_y, n * sizeof(float)));
	CHECK(bla_malloc((void**)&r->d_order, n * sizeof(int)));
	CHECK(bla_memcpy_h2d(r->d_X, store->X, n * N0 * sizeof(float), NULL));
	CHECK(bla_memcpy_h2d(r->d_y, store->y, n * sizeof(float), NULL));
	CHECK(bla_stream_sync(NULL));
}

static void open_store(MnistCSV* store, const char* path) {
	MnistCSV s = {fopen(path, "r"), NULL, NULL, 0, 0, NULL};
	if (!s.file) { fprintf(stderr, "cannot open %s\n", path); exit(1); }
	mnist_csv_init(&s);
	*store = s;
}

/* ---- train: model/mnist_nn.c:164-394 ------------------------------------------------------------------------------------------- */
static void train(int num_epochs, int batch, int colsum_mode) {
	float* flat = load_params();
	MnistCSV store;
	open_store(&store, env_or("BLA_MNIST_TRAIN", "data/mnist/mnist_train.csv"));
	const int N = store.num_examples;
	int gpus = atoi(env_or("BLA_GPUS", "1"));
	const int share = atoi(env_or("BLA_SHARE_GPU", "0"));        /* rehearsal: all replicas on device 0 */
	if (gpus < 1 || gpus > MAX_GPUS || batch % gpus) { fprintf(stderr, "BLA_GPUS=%d must divide the batch %d (max %d)\n", gpus, batch, MAX_GPUS); exit(1); }
	const int per = batch / gpus;
	const int num_batches = (int)ceil((float)N / (float)batch);     /* :187 */
	const int last = N - (num_batches - 1) * batch;                 /* :194-195: the final batch takes what is left */
	if (last != batch && last % gpus) { fprintf(stderr, "the last batch of an epoch (%d examples) does not divide over %d GPUs\n", last, gpus); exit(1); }
	CHECK(bla_init(0));
	Replica rep[MAX_GPUS];
	memset(rep, 0, sizeof rep);
	char handles[MAX_GPUS * BLA_DP_HANDLE_BYTES];
	for (int r = 0; r < gpus; r++) {
		if (gpus > 1) CHECK(bla_context_create(&rep[r].ctx, share ? 0 : r));
		use(&rep[r]);
		CHECK(bla_mnist_nn_create(&rep[r].nn, kSizes, per));
		CHECK(bla_mnist_nn_metrics_enable(rep[r].nn, 1));
		if (last != batch) {      /* adopting a bucket copies the adopter's current parameters into it: load the weights afterwards */
			CHECK(bla_mnist_nn_create(&rep[r].tail, kSizes, last / gpus));
			CHECK(bla_mnist_nn_use_buckets(rep[r].tail, bla_mnist_nn_params(rep[r].nn), bla_mnist_nn_grads(rep[r].nn)));
			CHECK(bla_mnist_nn_metrics_enable(rep[r].tail, 1));
		}
		CHECK(bla_mnist_nn_set_params(rep[r].nn, flat));
		upload_dataset(&rep[r], &store);
		if (gpus > 1) {
			CHECK(bla_dp_create(&rep[r].dp, r, gpus, bla_mnist_nn_param_count(rep[r].nn)));
			CHECK(bla_dp_export(rep[r].dp, handles + (size_t)r * BLA_DP_HANDLE_BYTES));
		}
	}
	for (int r = 0; r < gpus && gpus > 1; r++) { use(&rep[r]); CHECK(bla_dp_connect(rep[r].dp, handles)); }

	int* next_order = NULL;
	for (int i = 0; i < num_epochs; i++) {
		const float epoch_learn_rate = -SGD_LEARN_RATE_MULTIPLIER;       /* :186 (a float) */
		const double t0 = now_s();
		int* order = next_order;
		if (!order) {
			memset(store.sampled, 0, (size_t)N);                         /* :189-191 */
			store.num_sampled = 0;
			rng_begin();
			order = take_order(&store, N);                               /* the N draws of :205, in order */
			rng_end();
		}
		next_order = NULL;
		for (int r = 0; r < gpus; r++) { use(&rep[r]); CHECK(bla_memcpy_h2d(rep[r].d_order, order, (size_t)N * sizeof(int), NULL)); CHECK(bla_stream_sync(NULL)); }
		if (getenv("BLA_MNIST_SELFCHECK")) {     /* debugging aid: what the device holds against what the host sent */
			use(&rep[0]);
			float* back = malloc((size_t)N * N0 * sizeof(float));
			CHECK(bla_memcpy_d2h(back, rep[0].d_X, (size_t)N * N0 * sizeof(float), NULL)); CHECK(bla_stream_sync(NULL));
			fprintf(stderr, "[selfcheck] dataset %s\n", memcmp(back, store.X, (size_t)N * N0 * sizeof(float)) ? "DIFFERS" : "ok");
			int* ob = malloc((size_t)N * sizeof(int));
			CHECK(bla_memcpy_d2h(ob, rep[0].d_order, (size_t)N * sizeof(int), NULL)); CHECK(bla_stream_sync(NULL));
			fprintf(stderr, "[selfcheck] order %s, head %d %d %d %d\n", memcmp(ob, order, (size_t)N * sizeof(int)) ? "DIFFERS" : "ok", order[0], order[1], order[2], order[3]);
			float* pb = malloc(param_count() * sizeof(float));
			CHECK(bla_mnist_nn_get_params(rep[0].nn, pb));
			fprintf(stderr, "[selfcheck] params %s\n", i == 0 && memcmp(pb, flat, param_count() * sizeof(float)) ? "DIFFERS" : "ok");
			CHECK(bla_mnist_nn_gather_batch(rep[0].nn, NULL, rep[0].d_X, rep[0].d_y, N, rep[0].d_order));
			float* xb = malloc((size_t)N0 * per * sizeof(float));
			CHECK(bla_memcpy_d2h(xb, bla_mnist_nn_input(rep[0].nn), (size_t)N0 * per * sizeof(float), NULL)); CHECK(bla_stream_sync(NULL));
			int bad = 0;
			for (int p = 0; p < N0; p++) for (int k = 0; k < per && k < N; k++) bad += xb[(size_t)p * per + k] != store.X[(size_t)p * N + order[k]];
			fprintf(stderr, "[selfcheck] gathered batch: %d mismatches\n", bad);
			free(back); free(ob); free(pb); free(xb);
		}
		for (int j = 0; j < num_batches; j++) {
			const int in_this_batch = j == num_batches - 1 ? last : batch;
			for (int r = 0; r < gpus; r++) {       /* every replica's launches are queued before any of them is waited for */
				use(&rep[r]);
				bla_mnist_nn* nn = in_this_batch == batch ? rep[r].nn : rep[r].tail;
				const int mine = in_this_batch / gpus;
				CHECK(bla_mnist_nn_gather_batch(nn, NULL, rep[r].d_X, rep[r].d_y, N, rep[r].d_order + (size_t)j * batch + (size_t)r * mine));
				if (getenv("BLA_MNIST_SYNC_AFTER_GATHER")) CHECK(bla_stream_sync(NULL));
				if (gpus == 1) CHECK(bla_mnist_nn_fused_step(nn, NULL, NULL, NULL, epoch_learn_rate, colsum_mode));
				else CHECK(bla_mnist_nn_dp_step_direct(nn, rep[r].dp, NULL, epoch_learn_rate, colsum_mode));
			}
		}
		if (i + 1 < num_epochs) {      /* the next epoch's draws do not depend on this epoch's results: make them while the GPUs work */
			memset(store.sampled, 0, (size_t)N);
			store.num_sampled = 0;
			rng_begin();
			next_order = take_order(&store, N);
			rng_end();
		}
		double epoch_avg_loss = 0, epoch_avg_accuracy = 0;
		for (int r = 0; r < gpus; r++) {
			use(&rep[r]);
			double l; long long c;
			CHECK(bla_mnist_nn_metrics_read(rep[r].nn, &l, &c, 1));
			epoch_avg_loss += l; epoch_avg_accuracy += (double)c;
			if (rep[r].tail) { CHECK(bla_mnist_nn_metrics_read(rep[r].tail, &l, &c, 1)); epoch_avg_loss += l; epoch_avg_accuracy += (double)c; }
		}
		const double dt = now_s() - t0;
		epoch_avg_accuracy /= (float)N;                                  /* :339-340 */
		epoch_avg_loss /= (float)N;
		printf("Epoch %d:\tAvg accuracy: %.3f\tAvg loss: %.5f\n", i, epoch_avg_accuracy, epoch_avg_loss);
		fflush(stdout);
		fprintf(stderr, "[mnist_nn_gpu] epoch %d: %d examples in %.3f ms on %d GPU(s) = %.0f samples/s end to end (sampler, order upload, gather, step, metrics)\n", i, N,
		        dt * 1e3, gpus, N / dt);
		free(order);
	}
	use(&rep[0]);
	CHECK(bla_mnist_nn_get_params(rep[0].nn, flat));
	save_params(flat);
	for (int r = 0; r < gpus; r++) {
		use(&rep[r]);
		if (rep[r].dp) { int status = 0; CHECK(bla_dp_status(rep[r].dp, &status)); if (status) { fprintf(stderr, "exchange status %d on rank %d\n", status, r); exit(2); } }
	}
	for (int r = 0; r < gpus; r++) {
		use(&rep[r]);
		if (rep[r].dp) CHECK(bla_dp_destroy(rep[r].dp));
		if (rep[r].tail) CHECK(bla_mnist_nn_destroy(rep[r].tail));
		CHECK(bla_mnist_nn_destroy(rep[r].nn));
		CHECK(bla_free(rep[r].d_X)); CHECK(bla_free(rep[r].d_y)); CHECK(bla_free(rep[r].d_order));
	}
	free(flat); free(store.X); free(store.y); free(store.sampled);
}

/* ---- run: model/mnist_nn.c:401-510 ----------------------------------------------------------------------------------------------- */
static void run(int num_predictions) {
	float* flat = load_params();
	MnistCSV store;
	open_store(&store, env_or("BLA_MNIST_TEST", "data/mnist/mnist_test.csv"));
	if (num_predictions == -1 || num_predictions > store.num_examples) num_predictions = store.num_examples;     /* :421-423 */
	printf("Running predictions for %d digits...", num_predictions);
	fflush(stdout);
	rng_begin();
	int* order = take_order(&store, num_predictions);                     /* the draws of :435 */
	rng_end();
	CHECK(bla_init(0));
	Replica rep;
	memset(&rep, 0, sizeof rep);
	upload_dataset(&rep, &store);
	CHECK(bla_memcpy_h2d(rep.d_order, order, (size_t)num_predictions * sizeof(int), NULL));
	CHECK(bla_stream_sync(NULL));
	/* the reference runs ONE forward pass over all columns; columns are independent, so chunks give the same counts */
	const int chunk = 1024;
	long long num_correct = 0;
	bla_mnist_nn* nn = NULL;
	int width = 0;
	for (int done = 0; done < num_predictions || nn;) {
		const int w = num_predictions - done < chunk ? num_predictions - done : chunk;
		if (nn && w != width) {        /* the chunk width changes (last, shorter chunk) or everything is done: collect and release */
			double l; long long c;
			CHECK(bla_mnist_nn_metrics_read(nn, &l, &c, 1));
			num_correct += c;
			CHECK(bla_mnist_nn_destroy(nn));
			nn = NULL;
		}
		if (w == 0) break;
		if (!nn) {
			CHECK(bla_mnist_nn_create(&nn, kSizes, w));
			CHECK(bla_mnist_nn_set_params(nn, flat));
			CHECK(bla_mnist_nn_metrics_enable(nn, 1));
			width = w;
		}
		CHECK(bla_mnist_nn_gather_batch(nn, NULL, rep.d_X, rep.d_y, store.num_examples, rep.d_order + done));
		CHECK(bla_mnist_nn_forward(nn, NULL, NULL, NULL));
		done += w;
	}
	printf("done! Got %d correct (%.3f).\n", (int)num_correct, (float)num_correct / (float)num_predictions);     /* :490 */
	CHECK(bla_free(rep.d_X)); CHECK(bla_free(rep.d_y)); CHECK(bla_free(rep.d_order));
	free(order); free(flat); free(store.X); free(store.y); free(store.sampled);
}

int main(int argc, char** argv) {      /* model/mnist_nn.c:512-536, plus the batch size / col-sum mode arguments */
	rng_seed(42);                      /* srand(42), :513 */
	if (argc < 2) {
		printf("Please supply an argument, options:\n\trun [<num predictions>]\n\ttrain <num epochs> [<batch size> [as-written]]\n\tinit\n");
		exit(1);
	}
	if (strncmp(argv[1], "run", 3) == 0) {
		run(argc < 3 ? -1 : atoi(argv[2]));
	} else if (strncmp(argv[1], "train", 5) == 0) {
		if (argc < 3) {
			printf("Please supply a number of epochs, usage:\n\ttrain <num_epochs> [<batch size> [as-written]]\n");
			exit(1);
		}
		const int batch = argc > 3 ? atoi(argv[3]) : 64;                 /* SGD_BATCH_SIZE, :11 */
		const int mode = argc > 4 && strcmp(argv[4], "as-written") == 0 ? BLA_COLSUM_AS_WRITTEN : BLA_COLSUM_INTENDED;
		if (batch < 1) { printf("batch size must be positive\n"); exit(1); }
		train(atoi(argv[2]), batch, mode);
	} else if (strncmp(argv[1], "init", 4) == 0) {
		init();
	} else if (strcmp(argv[1], "order") == 0 && argc == 5) {
		/* self-check (tests/test_c_trainer.py): `order <csv> <draws> fast|walk` prints the example indices the sampler yields, by
		 * take_order() or by get_random_data_take()'s own walk -- the two lists must be identical */
		MnistCSV store;
		open_store(&store, argv[2]);
		const int draws = atoi(argv[3]);
		rng_begin();
		if (strcmp(argv[4], "fast") == 0) {
			int* order = take_order(&store, draws);
			for (int d = 0; d < draws; d++) printf("%d\n", order[d]);
			free(order);
		} else {
			for (int d = 0; d < draws; d++) printf("%d\n", (int)(get_random_data_take(&store).X - store.X));
		}
		rng_end();
	} else {
		printf("Unrecognized argument, options:\n\trun [<num predictions>]\n\ttrain <num epochs> [<batch size> [as-written]]\n\tinit\n");
		exit(1);
	}
	return 0;
}
