/* TEST INFRASTRUCTURE.  Runs the scalar C oracles (msda_cpu.c, point_sample_cpu.c) on one case
 * read from a file, every buffer a heap block of exactly the size the case implies, so that a
 * build with -fsanitize=address,undefined reports any out-of-bounds access or undefined
 * arithmetic of the oracle.  Used by tests/test_oracle.py on the golden vectors.
 *
 *   sanitize_driver <case file> <result file>
 *
 * Case file: int32 kind (0 attention, 1 point sampling), int32 is_f64, int32 dims[7], then the
 * inputs as raw arrays.  kind 0: dims = N S M D Lq L P; shapes (L*2 int64), lsi (L int64), value,
 * loc, attn, grad_out.  kind 1: dims = N C H W P mode align_corners; input, coords, grad_out.
 * Result file: the outputs as raw arrays, forward first. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

void msda_cpu_forward_f32(const float*, const int64_t*, const int64_t*, const float*, const float*, float*, int, int, int, int, int, int, int);
void msda_cpu_forward_f64(const double*, const int64_t*, const int64_t*, const double*, const double*, double*, int, int, int, int, int, int, int);
void msda_cpu_backward_f32(const float*, const float*, const int64_t*, const int64_t*, const float*, const float*, float*, float*, float*, int, int, int, int, int, int, int);
void msda_cpu_backward_f64(const double*, const double*, const int64_t*, const int64_t*, const double*, const double*, double*, double*, double*, int, int, int, int, int, int, int);
void point_sample_cpu_forward_f32(const float*, const float*, float*, int, int, int, int, int, int, int);
void point_sample_cpu_forward_f64(const double*, const double*, double*, int, int, int, int, int, int, int);
void point_sample_cpu_backward_f32(const float*, const float*, const float*, float*, float*, int, int, int, int, int, int, int);
void point_sample_cpu_backward_f64(const double*, const double*, const double*, double*, double*, int, int, int, int, int, int, int);

static void* read_block(FILE* f, size_t bytes) {
  void* p = malloc(bytes ? bytes : 1);
  if (!p || fread(p, 1, bytes, f) != bytes) {
    fprintf(stderr, "sanitize_driver: short case file\n");
    exit(2);
  }
  return p;
}

static void write_block(FILE* f, void* p, size_t bytes) {
  if (fwrite(p, 1, bytes, f) != bytes) exit(3);
  free(p);
}

int main(int argc, char** argv) {
  if (argc != 3) return 64;
  FILE* in = fopen(argv[1], "rb");
  FILE* out = fopen(argv[2], "wb");
  if (!in || !out) return 65;
  int32_t head[9];
  if (fread(head, sizeof(int32_t), 9, in) != 9) return 66;
  const int kind = head[0], f64 = head[1];
  const int* d = head + 2;
  const size_t e = f64 ? 8 : 4;
  if (kind == 0) {
    const size_t n = d[0], s = d[1], m = d[2], dd = d[3], q = d[4], l = d[5], p = d[6];
    int64_t* shapes = read_block(in, l * 2 * 8);
    int64_t* lsi = read_block(in, l * 8);
    void* value = read_block(in, n * s * m * dd * e);
    void* loc = read_block(in, n * q * m * l * p * 2 * e);
    void* attn = read_block(in, n * q * m * l * p * e);
    void* go = read_block(in, n * q * m * dd * e);
    void* o = malloc(n * q * m * dd * e);
    void* gv = malloc(n * s * m * dd * e);
    void* gl = malloc(n * q * m * l * p * 2 * e);
    void* ga = malloc(n * q * m * l * p * e);
    if (f64) {
      msda_cpu_forward_f64(value, shapes, lsi, loc, attn, o, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
      msda_cpu_backward_f64(go, value, shapes, lsi, loc, attn, gv, gl, ga, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    } else {
      msda_cpu_forward_f32(value, shapes, lsi, loc, attn, o, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
      msda_cpu_backward_f32(go, value, shapes, lsi, loc, attn, gv, gl, ga, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    }
    write_block(out, o, n * q * m * dd * e);
    write_block(out, gv, n * s * m * dd * e);
    write_block(out, gl, n * q * m * l * p * 2 * e);
    write_block(out, ga, n * q * m * l * p * e);
    free(shapes); free(lsi); free(value); free(loc); free(attn); free(go);
  } else {
    const size_t n = d[0], c = d[1], h = d[2], w = d[3], p = d[4];
    void* input = read_block(in, n * c * h * w * e);
    void* coords = read_block(in, n * p * 2 * e);
    void* go = read_block(in, n * c * p * e);
    void* o = malloc(n * c * p * e);
    void* gi = malloc(n * c * h * w * e);
    void* gc = malloc(n * p * 2 * e);
    if (f64) {
      point_sample_cpu_forward_f64(input, coords, o, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
      point_sample_cpu_backward_f64(go, input, coords, gi, gc, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    } else {
      point_sample_cpu_forward_f32(input, coords, o, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
      point_sample_cpu_backward_f32(go, input, coords, gi, gc, d[0], d[1], d[2], d[3], d[4], d[5], d[6]);
    }
    write_block(out, o, n * c * p * e);
    write_block(out, gi, n * c * h * w * e);
    write_block(out, gc, n * p * 2 * e);
    free(input); free(coords); free(go);
  }
  fclose(in);
  return fclose(out) == 0 ? 0 : 67;
}
