/* layer.c -- the reference's dense-layer API (lib/layer.c) on top of the drop-in matrix.h.  All arithmetic
 * goes through matrix_multiply / matrix_add / matrix_multiply_elementwise / matrix_scale / matrix_transpose, i.e.
 * through the device; operands here are vectors and rank-1 updates, so this path exists for API completeness,
 * not speed. */
#include "layer.h"
#include "matrix.h"
#include "csv.h"
#include <stdlib.h>

/* z = W a_prev + b ; a = act(z)   (reference lib/layer.c:6-20, including which structs it frees) */
void feed_forward(struct Layer* l) {
	if (!l->has_previous_layer) return;
	struct Matrix* z = matrix_multiply(*l->weights, *l->previous_layer->nodes);
	matrix_add(z, l->biases);
	if (l->has_nodes) {   /* the reference releases only the structs here, not their data (lib/layer.c:12-15) */
		free(l->raw_nodes);
		free(l->nodes);
	}
	l->raw_nodes = clone_matrix(*z);
	l->activation(z->data, l->num_nodes);
	l->nodes = z;
	l->has_nodes = 1;
}

void free_layer_data(struct Layer l) {                              /* reference lib/layer.c:22-32 */
	if (l.has_nodes) {
		free_matrix(l.raw_nodes);
		free_matrix(l.nodes);
	}
	if (l.has_previous_layer) {
		free_matrix(l.weights);
		free_matrix(l.biases);
	}
}

void load_weights_from_csv(struct Layer* l, const char* filepath) {  /* reference lib/layer.c:34-39 */
	if (l->has_previous_layer) l->weights = make_matrix(l->num_nodes, l->previous_layer->num_nodes, read_csv_contents(filepath));
}

void load_biases_from_csv(struct Layer* l, const char* filepath) {   /* reference lib/layer.c:41-46 */
	if (l->has_previous_layer) l->biases = make_matrix(l->num_nodes, 1, read_csv_contents(filepath));
}

/* delta = act'(z) (.) dC/da * (-learn_rate)  -- the bias step of layer l */
static struct Matrix* scaled_delta(struct Layer* l, struct Matrix* cost_ddx_activation, float learn_rate) {
	struct Matrix* d = clone_matrix(*l->raw_nodes);
	l->activation_ddx(d->data, l->num_nodes);
	matrix_multiply_elementwise(d, cost_ddx_activation);
	matrix_scale(d, -learn_rate);
	return d;
}

/* dW = delta . a_prev^T, evaluated the reference's way: transpose, multiply, transpose back */
static struct Matrix* outer_with_previous(struct Layer* l, struct Matrix* delta) {
	matrix_transpose(l->previous_layer->nodes);
	struct Matrix* dw = matrix_multiply(*delta, *l->previous_layer->nodes);
	matrix_transpose(l->previous_layer->nodes);
	return dw;
}

static void descend(struct Layer* l, struct Layer* next, struct Matrix* cost_ddx_next_activation, float learn_rate);

/* shared tail of both entry points: compute this layer's step, recurse toward the input with the OLD weights,
 * then apply the step (reference lib/layer.c:64-77 and :92-106 have the same order) */
static void step_and_recurse(struct Layer* l, struct Matrix* cost_ddx_activation, float learn_rate) {
	struct Matrix* db = scaled_delta(l, cost_ddx_activation, learn_rate);
	struct Matrix* dw = outer_with_previous(l, db);
	descend(l->previous_layer, l, cost_ddx_activation, learn_rate);
	matrix_add(l->weights, dw);
	matrix_add(l->biases, db);
	free_matrix(dw);
	free_matrix(db);
}

/* hidden layer: dC/da_l = W_next^T (act'_next(z_next) (.) dC/da_next)   (reference lib/layer.c:48-62) */
static void descend(struct Layer* l, struct Layer* next, struct Matrix* cost_ddx_next_activation, float learn_rate) {
	if (!l->has_previous_layer) return;
	struct Matrix* g = clone_matrix(*next->raw_nodes);
	next->activation_ddx(g->data, next->num_nodes);
	matrix_multiply_elementwise(g, cost_ddx_next_activation);
	matrix_transpose(next->weights);
	struct Matrix* cost_ddx_activation = matrix_multiply(*next->weights, *g);
	matrix_transpose(next->weights);
	free_matrix(g);
	step_and_recurse(l, cost_ddx_activation, learn_rate);
	free_matrix(cost_ddx_activation);
}

/* output layer: dC/da = 2 (a - y)   (reference lib/layer.c:80-90) */
void back_propagate_errors(struct Layer* l, float* expectations, float learn_rate) {
	if (!l->has_previous_layer) return;
	struct Matrix* cost_ddx_activation = clone_matrix(*l->nodes);
	for (int i = 0; i < l->num_nodes; i++) cost_ddx_activation->data[i] = 2 * (cost_ddx_activation->data[i] - expectations[i]);
	step_and_recurse(l, cost_ddx_activation, learn_rate);
	free_matrix(cost_ddx_activation);
}
