// TEST INFRASTRUCTURE ONLY -- never linked into, imported by or executed from the product path.
//
// Golden-vector generator: drives the *real* reference implementation, compiled straight from
// the headers where they lie under the reference checkout (include path given on the compiler
// command line; nothing from the reference is copied into this repository).  The binary lands
// in oracle/_ref/ (git-ignored).  It is used
//   * in the build container, to produce the fixtures under tests/golden/ (see
//     tools/make_golden.py) that pin both the CPU restatement in oracle/vtcnn_oracle.cpp and
//     the HIP path;
//   * on the GPU box, as the timed "reference CPU path" of bench.py (mode `bench`).
//
// Build (see oracle/Makefile):
//   g++ -std=c++17 -O2 -ffp-contract=off -fopenmp -I$REF -I$REF/utils -I$REF/utils/tensor ...
//
// Every dump is made with OMP_NUM_THREADS=1 so that the reference's thread-count dependent
// float reductions (tensor_cpu.hpp:489-520) are deterministic.
#include <bits/stdc++.h>
#include <omp.h>
#include <filesystem>

#define private public
#define protected public
#include "utils/load_dataset.hpp"
#include "utils/trainer.hpp"
#include "model/cnn.hpp"
#include "model/mlp.hpp"
#include "model/vit_cnn.hpp"
#undef private
#undef protected

std::mt19937 Tensor::global_gen(42);

// ---------------------------------------------------------------------------------------------
// Bundle writer: [i32 name_len][name][i32 dtype 0=f32 1=i32][i32 ndim][i32 dims...][payload]
// ---------------------------------------------------------------------------------------------
struct Bundle
{
  std::ofstream out;
  explicit Bundle(const std::string &path) : out(path, std::ios::binary)
  {
    if (!out)
      throw std::runtime_error("cannot open " + path);
  }
  void header(const std::string &name, int dtype, const std::vector<int> &dims)
  {
    int nl = (int)name.size();
    out.write((const char *)&nl, 4);
    out.write(name.data(), nl);
    out.write((const char *)&dtype, 4);
    int nd = (int)dims.size();
    out.write((const char *)&nd, 4);
    out.write((const char *)dims.data(), 4 * nd);
  }
  void put(const std::string &name, const Tensor &t)
  {
    std::vector<int> dims = t.shape;
    long total = 1;
    for (int d : dims)
      total *= d;
    if (total != (long)t.data.size())
      dims = {(int)t.data.size()};
    header(name, 0, dims);
    out.write((const char *)t.data.data(), 4 * t.data.size());
  }
  void putf(const std::string &name, const std::vector<float> &v, std::vector<int> dims = {})
  {
    if (dims.empty())
      dims = {(int)v.size()};
    header(name, 0, dims);
    out.write((const char *)v.data(), 4 * v.size());
  }
  void puti(const std::string &name, const std::vector<int> &v, std::vector<int> dims = {})
  {
    if (dims.empty())
      dims = {(int)v.size()};
    header(name, 1, dims);
    out.write((const char *)v.data(), 4 * v.size());
  }
};

struct Hyper
{
  int image_channels = 1, input_dim = 28, vt_layers = 1, tokens = 16, token_channels = 32;
  int vt_channels = 16, enc_layers = 2, heads = 2, fc_dim = 64, num_classes = 10;
};

// Hyper-parameters other than the class count come from the environment when set (the scaled and
// wide shapes): VT_REF_CI, VT_REF_H, VT_REF_TOKENS, VT_REF_D, VT_REF_C2, VT_REF_HEADS.
static void hyper_from_env(Hyper &h)
{
  auto geti = [](const char *k, int &v) { if (const char *e = std::getenv(k)) v = std::atoi(e); };
  geti("VT_REF_CI", h.image_channels);
  geti("VT_REF_H", h.input_dim);
  geti("VT_REF_TOKENS", h.tokens);
  geti("VT_REF_D", h.token_channels);
  geti("VT_REF_C2", h.vt_channels);
  geti("VT_REF_HEADS", h.heads);
}

static VTCNN *make_model(const Hyper &h)
{
  Tensor::set_seed(42);
  return new VTCNN(h.image_channels, h.input_dim, h.vt_layers, h.tokens, h.token_channels, h.vt_channels,
                   h.enc_layers, h.heads, h.fc_dim, 0.1f, h.num_classes);
}

static void put_dense(Bundle &b, const std::string &pfx, DenseLayer *d, bool grads)
{
  if (!grads)
  {
    b.put(pfx + ".W", d->weights);
    if (d->use_bias)
      b.put(pfx + ".b", d->bias);
  }
  else
  {
    b.put(pfx + ".W", d->grad_weights);
    if (d->use_bias)
      b.put(pfx + ".b", d->grad_bias);
  }
}

template <class BN>
static void put_bn(Bundle &b, const std::string &pfx, BN *n, bool grads)
{
  if (!grads)
  {
    b.put(pfx + ".gamma", n->gamma);
    b.put(pfx + ".beta", n->beta);
    b.put(pfx + ".running_mean", n->running_mean);
    b.put(pfx + ".running_var", n->running_var);
  }
  else
  {
    b.put(pfx + ".gamma", n->grad_gamma);
    b.put(pfx + ".beta", n->grad_beta);
  }
}

// Dumps every parameter (grads=false) or every gradient (grads=true) under `<tag>.<tensor>`.
static void put_all(Bundle &b, const std::string &tag, VTCNN &m, bool grads)
{
  VisualTransformer *vt = m.vt_layers[0];
  b.put(tag + ".conv.W", grads ? m.conv_layer->grad_weights : m.conv_layer->weights);
  b.put(tag + ".conv.b", grads ? m.conv_layer->grad_biases : m.conv_layer->biases);
  put_bn(b, tag + ".bn", m.bn, grads);
  put_dense(b, tag + ".tok.linear1", vt->tokenizer->linear1, grads);
  put_dense(b, tag + ".tok.linear2", vt->tokenizer->linear2, grads);
  MultiHeadAttention *a = vt->transformer->attention;
  for (int h = 0; h < a->num_heads; ++h)
  {
    put_dense(b, tag + ".mha.q" + std::to_string(h), a->query_linears[h], grads);
    put_dense(b, tag + ".mha.k" + std::to_string(h), a->key_linears[h], grads);
    put_dense(b, tag + ".mha.v" + std::to_string(h), a->value_linears[h], grads);
  }
  put_dense(b, tag + ".mha.o", a->output_linear, grads);
  put_dense(b, tag + ".ffn.linear1", vt->transformer->linear1, grads);
  put_dense(b, tag + ".ffn.linear2", vt->transformer->linear2, grads);
  put_bn(b, tag + ".norm1", vt->transformer->norm1, grads);
  put_bn(b, tag + ".norm2", vt->transformer->norm2, grads);
  put_dense(b, tag + ".proj.linear1", vt->projector->linear1, grads);
  put_dense(b, tag + ".proj.linear2", vt->projector->linear2, grads);
  put_dense(b, tag + ".proj.linear3", vt->projector->linear3, grads);
  put_bn(b, tag + ".proj.norm", vt->projector->norm, grads);
  put_dense(b, tag + ".proj.downsample", vt->projector->downsample, grads);
  put_dense(b, tag + ".fc", m.fc, grads);
}

// Synthetic batch: pixels are u8 values from mt19937(seed) divided by 255.0f exactly like the IDX
// reader does (load_dataset.hpp:50); labels uniform in [0, K).
static void synth_batch(int N, int C, int HW, int K, unsigned seed, std::vector<float> &x, std::vector<int> &y)
{
  std::mt19937 g(seed);
  x.resize((size_t)N * C * HW);
  y.resize(N);
  for (auto &v : x)
    v = static_cast<float>(g() & 0xFFu) / 255.0f;
  for (auto &v : y)
    v = (int)(g() % (unsigned)K);
}

static int cmd_step(int argc, char **argv)
{
  // step <out.bundle> <N> <K> <seed> <full:0|1>
  std::string path = argv[2];
  int N = std::atoi(argv[3]);
  Hyper h;
  hyper_from_env(h);
  h.num_classes = std::atoi(argv[4]);
  unsigned seed = (unsigned)std::atoi(argv[5]);
  bool full = std::atoi(argv[6]) != 0;
  const int K = h.num_classes;

  std::unique_ptr<VTCNN> mp(make_model(h));
  VTCNN &m = *mp;
  VisualTransformer *vt = m.vt_layers[0];
  Bundle b(path);

  std::vector<float> xs;
  std::vector<int> ys;
  synth_batch(N, h.image_channels, h.input_dim * h.input_dim, K, seed, xs, ys);
  Tensor X({N, h.image_channels, h.input_dim, h.input_dim}, xs);
  std::vector<float> yf(ys.begin(), ys.end());
  Tensor y({N, 1}, yf);
  Tensor yoh = Tensor::one_hot(y, K);

  b.put("in.X", X);
  b.puti("in.labels", ys);
  put_all(b, "p0", m, false);

  Tensor preds = m.forward(X);
  b.put("fwd.preds", preds);
  b.putf("fwd.loss", {Tensor::compute_loss(preds, yoh)});
  b.puti("fwd.argmax", Tensor::argmax_batch(preds));
  b.put("fwd.logits", m.fc->output);
  if (full)
  {
    b.put("fwd.conv_out", m.conv_layer->output);
    b.put("fwd.x", vt->input_cache);               // [N,P,C1] BN2D output, token-major
    b.put("fwd.tok_A", vt->tokenizer->cache2);     // [N,P,L]
    b.put("fwd.tok_out", vt->tokenizer->token_cache);
    b.put("fwd.tr_a", vt->transformer->linear1->input);    // norm1 output
    b.put("fwd.tr_hdn", vt->transformer->linear1->output); // pre-relu
    b.put("fwd.tr_out", vt->token_cache);                  // transformer output relabelled [N,L,D]
    b.put("fwd.proj_a", vt->projector->cache);
    b.put("fwd.proj_z", vt->projector->norm->input);
    b.put("fwd.proj_out", m.avgpool->input);
    b.put("fwd.pooled", m.avgpool->output);
    // MHA forward output (expected identically zero)
    Tensor mha = vt->transformer->attention->forward({vt->tokenizer->token_cache.permute({1, 0, 2})})[0];
    b.put("fwd.mha_out", mha);
  }

  m.backward(yoh);
  put_all(b, "g", m, true);
  if (full)
  {
    b.put("bwd.d_logits", m.softmax->input_deltas);
    b.put("bwd.d_pooled", m.fc->input_deltas);
    b.put("bwd.d_projout", m.avgpool->input_deltas);
    b.put("bwd.proj_dz", vt->projector->norm->input_deltas);
    b.put("bwd.proj_dx", vt->projector->input_deltas);
    b.put("bwd.proj_dt", vt->projector->token_deltas);
    b.put("bwd.tr_ds2", vt->transformer->norm2->input_deltas);
    b.put("bwd.tr_dr", vt->transformer->norm1->input_deltas);
    b.put("bwd.tr_dx", vt->transformer->last_output);
    b.put("bwd.tok_dx", vt->tokenizer->input_deltas);
    b.put("bwd.bn_dx", m.bn->input_deltas);
  }

  m.update_weights(32.0f);
  m.zero_grad();
  put_all(b, "p1", m, false);

  Tensor preds2 = m.forward(X);
  b.put("fwd2.preds", preds2);
  m.set_training(false);
  Tensor pe = m.forward(X);
  b.put("eval.preds", pe);
  b.puti("eval.argmax", Tensor::argmax_batch(pe));
  put_all(b, "p2", m, false); // running statistics after the two extra forwards
  return 0;
}

// Data-parallel oracle: G independent mini-batches from identical weights, gradients summed on
// the host, one SGD step with divisor G*32.
static int cmd_dp(int argc, char **argv)
{
  // dp <out.bundle> <G> <N> <K> <seed>
  std::string path = argv[2];
  int G = std::atoi(argv[3]);
  int N = std::atoi(argv[4]);
  Hyper h;
  h.num_classes = std::atoi(argv[5]);
  unsigned seed = (unsigned)std::atoi(argv[6]);
  Bundle b(path);
  for (int g = 0; g < G; ++g)
  {
    std::unique_ptr<VTCNN> mp(make_model(h));
    std::vector<float> xs;
    std::vector<int> ys;
    synth_batch(N, 1, 784, h.num_classes, seed + g, xs, ys);
    Tensor X({N, 1, 28, 28}, xs);
    std::vector<float> yf(ys.begin(), ys.end());
    Tensor yoh = Tensor::one_hot(Tensor({N, 1}, yf), h.num_classes);
    b.put("in" + std::to_string(g) + ".X", X);
    b.puti("in" + std::to_string(g) + ".labels", ys);
    Tensor preds = mp->forward(X);
    b.put("fwd" + std::to_string(g) + ".preds", preds);
    mp->backward(yoh);
    put_all(b, "g" + std::to_string(g), *mp, true);
  }
  return 0;
}

static int cmd_kat(int argc, char **argv)
{
  // kat <out.bundle> : shuffle indices, weight init, checkpoint bytes
  Bundle b(argv[2]);
  for (int n : {48000, 12000, 10000, 3421, 100, 33})
  {
    std::vector<std::vector<float>> imgs(n, std::vector<float>(1, 0.f));
    std::vector<int> lbls(n, 0);
    DataLoader dl(imgs, lbls, 32);
    std::vector<int> a(dl.indices.begin(), dl.indices.end());
    dl.reset();
    std::vector<int> c(dl.indices.begin(), dl.indices.end());
    if (n > 4000)
    {
      a.resize(64);
      c.resize(64);
    }
    b.puti("shuffle." + std::to_string(n) + ".ctor", a);
    b.puti("shuffle." + std::to_string(n) + ".reset1", c);
  }
  for (int K : {10, 8})
  {
    Hyper h;
    h.num_classes = K;
    std::unique_ptr<VTCNN> mp(make_model(h));
    std::string f = std::string(argv[2]) + ".init" + std::to_string(K) + ".bin";
    mp->save(f);
    if (K == 8)
      put_all(b, "init8", *mp, false);
  }
  return 0;
}

// KAT on real data: shipped 8-class checkpoint over the shipped BloodMNIST test split.
static int cmd_blood(int argc, char **argv)
{
  // blood <out.bundle> <images-idx> <labels-idx> <ckpt.bin> <max_images>
  Bundle b(argv[2]);
  int maxn = std::atoi(argv[6]);
  Dataset ds = load_dataset(argv[3], argv[4], maxn);
  Hyper h;
  h.num_classes = 8;
  std::unique_ptr<VTCNN> mp(make_model(h));
  {
    std::streambuf *old = std::cout.rdbuf();
    std::ostringstream sink;
    std::cout.rdbuf(sink.rdbuf());
    mp->load(argv[5]);
    std::cout.rdbuf(old);
  }
  mp->set_training(false);
  DataLoader dl(ds.images, ds.labels, 32);
  dl.reset(); // evaluate_test() resets before iterating (trainer.hpp:99)
  std::vector<int> order(dl.indices.begin(), dl.indices.end());
  std::vector<int> pred, truth;
  std::vector<float> allp;
  float loss = 0, acc = 0;
  int batches = 0;
  while (dl.has_next())
  {
    auto [X, y] = dl.next_batch();
    Tensor yoh = Tensor::one_hot(y, 8);
    Tensor p = mp->forward(X);
    loss += Tensor::compute_loss(p, yoh);
    acc += Tensor::compute_accuracy(p, y, 8);
    ++batches;
    auto pr = Tensor::argmax_batch(p);
    auto tr = y.to_vector();
    pred.insert(pred.end(), pr.begin(), pr.end());
    truth.insert(truth.end(), tr.begin(), tr.end());
    allp.insert(allp.end(), p.data.begin(), p.data.end());
  }
  b.puti("order", order);
  b.puti("pred", pred);
  b.puti("truth", truth);
  b.putf("probs", allp, {(int)pred.size(), 8});
  b.putf("loss", {loss / batches});
  b.putf("acc", {acc / batches});
  std::printf("blood: n=%zu loss=%.6f acc=%.4f%%\n", pred.size(), loss / batches, 100.f * acc / batches);
  return 0;
}

// Timed reference training steps on synthetic BloodMNIST-shaped data.
static int cmd_bench(int argc, char **argv)
{
  // bench <N> <K> <steps> <warmup>
  int N = std::atoi(argv[2]);
  Hyper h;
  h.num_classes = std::atoi(argv[3]);
  int steps = std::atoi(argv[4]), warm = std::atoi(argv[5]);
  std::unique_ptr<VTCNN> mp(make_model(h));
  std::vector<float> xs;
  std::vector<int> ys;
  synth_batch(N, 1, 784, h.num_classes, 123, xs, ys);
  Tensor X({N, 1, 28, 28}, xs);
  std::vector<float> yf(ys.begin(), ys.end());
  Tensor yoh = Tensor::one_hot(Tensor({N, 1}, yf), h.num_classes);
  double t0 = 0;
  for (int s = 0; s < warm + steps; ++s)
  {
    if (s == warm)
      t0 = omp_get_wtime();
    Tensor p = mp->forward(X);
    mp->backward(yoh);
    mp->update_weights(32.0f);
    mp->zero_grad();
  }
  double dt = omp_get_wtime() - t0;
  std::printf("{\"images_per_sec\": %.3f, \"threads\": %d, \"batch\": %d, \"steps\": %d, \"seconds\": %.3f}\n",
              steps * (double)N / dt, omp_get_max_threads(), N, steps, dt);
  return 0;
}

// The reference's custom-image loader (utils/load_image_stb.hpp:9-26) on each file of a folder, in
// the order load_custom_images_from_folder visits them (utils/load_dataset.hpp:167-209).
static int cmd_images(int argc, char **argv)
{
  // images <folder> <out.bin>: per accepted file 784 floats, in order; prints "<n> <name>..." to stdout
  Dataset ds = load_custom_images_from_folder(argv[2]);
  std::ofstream f(argv[3], std::ios::binary);
  for (auto &img : ds.images)
    f.write(reinterpret_cast<const char *>(img.data()), (std::streamsize)(img.size() * sizeof(float)));
  std::printf("images: %zu\n", ds.images.size());
  return 0;
}

// traj <out.bundle> <N> <K> <seed> <steps>: `steps` training steps (forward, backward,
// update_weights(32), zero_grad: utils/trainer.hpp:41-56) on synthetic batches seeded seed, seed+1, ...
// from the seed-42 model; dumps the loss and arg-max labels of every step and the final weights.
static int cmd_traj(int argc, char **argv)
{
  const int N = std::atoi(argv[3]);
  Hyper h;
  hyper_from_env(h);
  h.num_classes = std::atoi(argv[4]);
  const unsigned seed = (unsigned)std::atoi(argv[5]);
  const int steps = std::atoi(argv[6]);
  const int K = h.num_classes;
  std::unique_ptr<VTCNN> mp(make_model(h));
  VTCNN &m = *mp;
  Bundle b(argv[2]);
  put_all(b, "p0", m, false);
  std::vector<float> losses;
  std::vector<int> labels_all;
  for (int t = 0; t < steps; ++t)
  {
    std::vector<float> xs;
    std::vector<int> ys;
    synth_batch(N, h.image_channels, h.input_dim * h.input_dim, K, seed + (unsigned)t, xs, ys);
    Tensor X({N, h.image_channels, h.input_dim, h.input_dim}, xs);
    std::vector<float> yf(ys.begin(), ys.end());
    Tensor yoh = Tensor::one_hot(Tensor({N, 1}, yf), K);
    b.put("traj.x" + std::to_string(t), X);
    b.puti("traj.y" + std::to_string(t), ys);
    Tensor preds = m.forward(X);
    losses.push_back(Tensor::compute_loss(preds, yoh));
    for (int a : Tensor::argmax_batch(preds))
      labels_all.push_back(a);
    m.backward(yoh);
    m.update_weights(32.0f);
    m.zero_grad();
  }
  b.putf("traj.loss", losses);
  b.puti("traj.argmax", labels_all, {steps, N});
  put_all(b, "pT", m, false);
  return 0;
}

// mha <out.bundle> <B> <T> <D> <heads> <seed>: MultiHeadAttention::forward
// (layers/multi_head_attention.hpp:64-107) on a uniform random input.  Dumps the layer's own output
// (zero: Tensor::concat leaves its result unfilled) and, evaluated with the same reference Tensor and
// DenseLayer operations the layer runs (:79-92), every head's output, plus what the output projection
// gives for their channel-wise concatenation (the reading the layer's comments intend).
static int cmd_mha(int argc, char **argv)
{
  const int B = std::atoi(argv[3]), T = std::atoi(argv[4]), D = std::atoi(argv[5]), H = std::atoi(argv[6]);
  const unsigned seed = (unsigned)std::atoi(argv[7]);
  Tensor::set_seed(42);
  MultiHeadAttention mha(D, D, H);
  const int hd = D / H;
  std::mt19937 gen(seed);
  std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
  Tensor x({B, T, D});
  for (auto &v : x.data)
    v = dist(gen);
  Bundle b(argv[2]);
  b.put("x", x);
  b.put("out", mha.forward({x})[0]);
  Tensor x_flat = x.reshape({B * T, D});
  Tensor cat({B, T, D});
  for (int h = 0; h < H; ++h)
  {
    put_dense(b, "q" + std::to_string(h), mha.query_linears[h], false);
    put_dense(b, "k" + std::to_string(h), mha.key_linears[h], false);
    put_dense(b, "v" + std::to_string(h), mha.value_linears[h], false);
    Tensor query = mha.query_linears[h]->forward({x_flat})[0].reshape({B, T, hd});
    Tensor key = mha.key_linears[h]->forward({x_flat})[0].reshape({B, T, hd});
    Tensor value = mha.value_linears[h]->forward({x_flat})[0].reshape({B, T, hd});
    Tensor scores = key.matmul(query.transpose({0, 2, 1}));
    scores = scores * (1.0f / std::sqrt(static_cast<float>(hd)));
    Tensor head_out = scores.softmax(2).matmul(value);
    b.put("head" + std::to_string(h), head_out);
    for (int r = 0; r < B * T; ++r)
      for (int e = 0; e < hd; ++e)
        cat.data[(size_t)r * D + h * hd + e] = head_out.data[(size_t)r * hd + e];
  }
  put_dense(b, "o", mha.output_linear, false);
  b.put("out_concat", mha.output_linear->forward({cat.reshape({B * T, D})})[0].reshape({B, T, D}));
  return 0;
}

// selfattn <out.bundle> <B> <T> <D> <Dk> <seed>: SelfAttention::forward / backward
// (layers/self_attention.hpp:31-136) on a uniform random input and delta; dumps the two weight
// matrices, the output, the weight gradients and the input delta.
static int cmd_selfattn(int argc, char **argv)
{
  const int B = std::atoi(argv[3]), T = std::atoi(argv[4]), D = std::atoi(argv[5]), Dk = std::atoi(argv[6]);
  const unsigned seed = (unsigned)std::atoi(argv[7]);
  Tensor::set_seed(42);
  SelfAttention sa(D, Dk);
  std::mt19937 gen(seed);
  std::uniform_real_distribution<float> dist(-1.0f, 1.0f);
  Tensor x({B, T, D}), delta({B, T, D});
  for (auto &v : x.data)
    v = dist(gen);
  for (auto &v : delta.data)
    v = dist(gen);
  Tensor out = sa.forward({x})[0];
  sa.backward(&delta);
  Bundle b(argv[2]);
  b.put("x", x);
  b.put("delta", delta);
  b.put("key.W", sa.key_linear->weights);
  b.put("query.W", sa.query_linear->weights);
  b.put("out", out);
  b.put("g.key.W", sa.key_linear->grad_weights);
  b.put("g.query.W", sa.query_linear->grad_weights);
  b.put("dx", sa.input_deltas);
  return 0;
}

// models <out.bundle> <N> <seed>: one training step (forward, backward, update_weights(32), zero_grad:
// utils/trainer.hpp:41-56) of the reference's two other models, CNN (model/cnn.hpp:13-119) and MLP (model/mlp.hpp:13-178,
// with its DropoutLayer, layers/dropout_layer.hpp:4-52), each constructed right after Tensor::set_seed(42), on one
// synthetic batch; dumps the weights before and after, the gradients, the training-mode probabilities of the step,
// those of a second training-mode forward (for the MLP a second dropout mask from the generator as it then stands)
// and the eval-mode probabilities.
template <typename M>
static void model_step(Bundle &b, const std::string &pfx, M &m, const Tensor &X, const Tensor &yoh)
{
  auto dump = [&](const std::string &tag, bool grads) {
    int li = 0;
    for (Layer *l : m.layers)
    {
      const std::string n = pfx + "." + tag + ".l" + std::to_string(li++);
      if (auto *d = dynamic_cast<DenseLayer *>(l))
      {
        b.put(n + ".W", grads ? d->grad_weights : d->weights);
        b.put(n + ".b", grads ? d->grad_bias : d->bias);
      }
      else if (auto *c = dynamic_cast<Conv2DLayer *>(l))
      {
        b.put(n + ".W", grads ? c->grad_weights : c->weights);
        b.put(n + ".b", grads ? c->grad_biases : c->biases);
      }
      else if (auto *n2 = dynamic_cast<BatchNorm2DLayer *>(l))
      {
        b.put(n + ".gamma", grads ? n2->grad_gamma : n2->gamma);
        b.put(n + ".beta", grads ? n2->grad_beta : n2->beta);
        if (!grads)
        {
          b.put(n + ".running_mean", n2->running_mean);
          b.put(n + ".running_var", n2->running_var);
        }
      }
    }
  };
  dump("p0", false);
  b.put(pfx + ".preds", m.forward(X));
  m.backward(yoh);
  dump("g", true);
  m.update_weights(32.0f);
  m.zero_grad();
  dump("p1", false);
  b.put(pfx + ".preds2", m.forward(X));
  m.set_training(false);
  b.put(pfx + ".preds_eval", m.forward(X));
  m.set_training(true);
}

static int cmd_models(int argc, char **argv)
{
  const int N = std::atoi(argv[3]);
  const unsigned seed = (unsigned)std::atoi(argv[4]);
  Bundle b(argv[2]);
  for (int which = 0; which < 2; ++which)
  {
    const int K = which == 0 ? 10 : 8;
    std::vector<float> xs;
    std::vector<int> ys;
    synth_batch(N, 1, 784, K, seed + (unsigned)which, xs, ys);
    Tensor X({N, 1, 28, 28}, xs);
    std::vector<float> yf(ys.begin(), ys.end());
    Tensor yoh = Tensor::one_hot(Tensor({N, 1}, yf), K);
    const std::string pfx = which == 0 ? "cnn" : "mlp";
    b.put(pfx + ".X", X);
    b.puti(pfx + ".labels", ys);
    Tensor::set_seed(42);
    if (which == 0)
    {
      CNN m;
      model_step(b, pfx, m, X, yoh);
    }
    else
    {
      MLP m;
      model_step(b, pfx, m, X, yoh);
    }
  }
  return 0;
}

int main(int argc, char **argv)
{
  if (argc < 2)
  {
    std::fprintf(stderr, "usage: ref_harness step|dp|kat|blood|bench|images|traj|mha|selfattn|models ...\n");
    return 2;
  }
  std::string c = argv[1];
  try
  {
    if (c == "step" && argc == 7)
      return cmd_step(argc, argv);
    if (c == "dp" && argc == 7)
      return cmd_dp(argc, argv);
    if (c == "kat" && argc == 3)
      return cmd_kat(argc, argv);
    if (c == "blood" && argc == 7)
      return cmd_blood(argc, argv);
    if (c == "bench" && argc == 6)
      return cmd_bench(argc, argv);
    if (c == "images" && argc == 4)
      return cmd_images(argc, argv);
    if (c == "mha" && argc == 8)
      return cmd_mha(argc, argv);
    if (c == "traj" && argc == 7)
      return cmd_traj(argc, argv);
    if (c == "selfattn" && argc == 8)
      return cmd_selfattn(argc, argv);
    if (c == "models" && argc == 5)
      return cmd_models(argc, argv);
  }
  catch (const std::exception &e)
  {
    std::fprintf(stderr, "ref_harness: %s\n", e.what());
    return 1;
  }
  std::fprintf(stderr, "ref_harness: bad arguments\n");
  return 2;
}
