#ifndef __layer_h__
#define __layer_h__

/* Drop-in for the reference's lib/layer.h: a singly linked chain of dense layers working on n x 1 column
 * vectors, with the weight update folded into back-propagation.  Activation callbacks take float* -- which is
 * exactly matrix_float_t* in this build (see matrix.h). */
struct Layer {
	int num_nodes;
	struct Matrix* nodes;      /* activations a = act(z), n x 1 */
	struct Matrix* raw_nodes;  /* pre-activations z = W a_prev + b */
	struct Matrix* weights;    /* n x n_prev */
	struct Matrix* biases;     /* n x 1 */
	struct Layer* previous_layer;
	void (*activation)(float*, int);
	void (*activation_ddx)(float*, int);
	char has_previous_layer;
	char has_nodes;
};

void feed_forward(struct Layer* l);
void free_layer_data(struct Layer l);
void load_weights_from_csv(struct Layer* l, const char* filepath);
void load_biases_from_csv(struct Layer* l, const char* filepath);
void back_propagate_errors(struct Layer* l, float* expectations, float learn_rate);

#endif
