/* layer.h -- drop-in for the reference's lib/layer.h: a singly linked chain of dense layers on n x 1 column vectors
 * (the pre-batch API of my_first_model.c / mnist_hinge.c), with the weight update folded into back-propagation.
 *
 * `struct Layer` IS the interface: programs build the chain by filling its fields, so names, types and order are the
 * reference's (lib/layer.h:4-15).  The activation callbacks take float* -- which is exactly matrix_float_t* in this
 * build (matrix.h), and the only typedef under which the reference's own layer.c is type-correct.
 * The products inside run on the device through matrix.h's entry points. */
#ifndef BLA_DROPIN_LAYER_H
#define BLA_DROPIN_LAYER_H

struct Layer {
	int num_nodes;                         /* n */
	struct Matrix* nodes;                  /* activations a = act(z), n x 1 */
	struct Matrix* raw_nodes;              /* pre-activations z = W a_prev + b, n x 1 */
	struct Matrix* weights;                /* n x n_prev */
	struct Matrix* biases;                 /* n x 1 */
	struct Layer* previous_layer;          /* towards the input */
	void (*activation)(float*, int);       /* in place over n values */
	void (*activation_ddx)(float*, int);   /* derivative, in place */
	char has_previous_layer;               /* 0 for the input layer */
	char has_nodes;                        /* nodes / raw_nodes currently allocated */
};

/* z = W a_prev + b, a = act(z) for `layer` (its predecessors must be up to date); no-op on the input layer (lib/layer.c:6-20) */
void feed_forward(struct Layer* layer);

/* One SGD step from the output layer down: g = 2 (a - target), delta = act'(z) (.) g * (-rate), dW = delta a_prev^T, recurse,
 * then W += dW and b += delta (:48-107) */
void back_propagate_errors(struct Layer* output_layer, float* target, float rate);

/* weights / biases from comma-terminated CSV files of the layer's own shape (:35-46) */
void load_weights_from_csv(struct Layer* layer, const char* path);
void load_biases_from_csv(struct Layer* layer, const char* path);

/* releases nodes, raw_nodes, weights and biases (structs and data) of a layer passed BY VALUE (:22-33) */
void free_layer_data(struct Layer layer);

#endif /* BLA_DROPIN_LAYER_H */
