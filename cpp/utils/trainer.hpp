// Training / evaluation loop with the reference's semantics and console output
// (utils/trainer.hpp:9-243): per batch forward, backward, update_weights(batch_size), zero_grad;
// evaluation toggles set_training(false/true); evaluate_test prints macro/weighted F1, per-class
// precision/recall/F1 and the confusion matrix.
#pragma once
#include <chrono>
#include <iomanip>
#include <iostream>

#include <cstdlib>
#include <exception>
#include <map>
#include <memory>
#include <thread>

#include "../model/model.hpp"
#include "../model/vit_cnn.hpp"
#include "load_dataset.hpp"
#include "tensor.hpp"

class Trainer {
  Model &model;
  DataLoader &train_loader;
  DataLoader &val_loader;
  DataLoader *test_loader;
  int num_classes;
  int batch_size;

  struct Pass {
    float loss = 0.f, acc = 0.f;
    int batches = 0;
    std::vector<int> truth, pred;
  };
  // ---- device-resident path: taken when the model is the HIP VTCNN and the loader's dataset can live on the
  // device (set VTCNN_HOST_LOOP=1 to force the host loop below, which is the reference's own).  Same batches
  // in the same order, same arithmetic per batch; the per-batch loss and accuracy stay on the device until
  // the sweep ends, so nothing in the loop waits for the GPU.
  struct DeviceCounters {
    float *loss_sums = nullptr;
    int *correct = nullptr, *confusion = nullptr, *argmax = nullptr;
    int slots = 0, K = 0, images = 0;
    DeviceCounters(int slots_, int K_, int images_) : slots(slots_), K(K_), images(images_) {
      vt_host::hip_check(hipMalloc((void **)&loss_sums, sizeof(float) * slots), "hipMalloc");
      vt_host::hip_check(hipMalloc((void **)&correct, sizeof(int) * slots), "hipMalloc");
      vt_host::hip_check(hipMalloc((void **)&confusion, sizeof(int) * K * K), "hipMalloc");
      vt_host::hip_check(hipMalloc((void **)&argmax, sizeof(int) * std::max(images, 1)), "hipMalloc");
      vt_host::hip_check(hipMemset(confusion, 0, sizeof(int) * K * K), "hipMemset");
    }
    ~DeviceCounters() {
      (void)hipFree(loss_sums);
      (void)hipFree(correct);
      (void)hipFree(confusion);
      (void)hipFree(argmax);
    }
    DeviceCounters(const DeviceCounters &) = delete;
    DeviceCounters &operator=(const DeviceCounters &) = delete;
  };
  VTCNN *device_model(DataLoader &loader) {
    const char *force = std::getenv("VTCNN_HOST_LOOP");
    if (force && *force && *force != '0') return nullptr;
    VTCNN *m = dynamic_cast<VTCNN *>(&model);
    if (!m || (int)loader.batch() > m->max_batch() || !loader.make_resident()) return nullptr;
    return m;
  }
  // per-batch means summed in batch order, as the reference's loop accumulates them (utils/trainer.hpp:55-58)
  static void fold_batches(const std::vector<float> &loss_sums, const std::vector<int> &correct, size_t total, size_t bs, Pass &p) {
    for (size_t b = 0; b < loss_sums.size(); ++b) {
      const float n = (float)std::min(bs, total - b * bs);
      p.loss += loss_sums[b] / n;
      p.acc += static_cast<float>(correct[b]) / n;
      ++p.batches;
    }
  }
  Pass sweep_resident(VTCNN &m, DataLoader &loader, bool keep_labels) {
    Pass p;
    const size_t total = loader.size(), bs = loader.batch(), nb = loader.total_batches();
    DeviceCounters dc((int)nb, num_classes, (int)total);
    const DataLoader::Resident &r = loader.resident(m.stream());
    m.set_training(false);
    // consecutive loader batches go through one launch set per chunk; each keeps its own BatchNorm1D statistics
    const size_t per_chunk = (bs % 4 == 0) ? std::max<size_t>(1, (size_t)m.max_batch() / bs) : 1;
    for (size_t b0 = 0; b0 < nb; b0 += per_chunk) {
      const size_t first = b0 * bs, n = std::min(per_chunk * bs, total - first);
      m.eval_chunk_resident(r.images_u8, r.labels, r.indices + first, (int)n, (int)bs, (int)b0, dc.loss_sums, dc.correct,
                            dc.confusion, dc.argmax + first);
    }
    m.set_training(true);
    std::vector<float> ls(nb);
    std::vector<int> ok(nb);
    vt_host::hip_check(hipMemcpyAsync(ls.data(), dc.loss_sums, sizeof(float) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    vt_host::hip_check(hipMemcpyAsync(ok.data(), dc.correct, sizeof(int) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    if (keep_labels) {
      p.pred.resize(total);
      vt_host::hip_check(hipMemcpyAsync(p.pred.data(), dc.argmax, sizeof(int) * total, hipMemcpyDeviceToHost, m.stream()), "D2H");
      p.truth.resize(total);
      for (size_t i = 0; i < total; ++i) p.truth[i] = loader.all_labels()[loader.order()[i]];
    }
    vt_host::hip_check(hipStreamSynchronize(m.stream()), "sync");
    fold_batches(ls, ok, total, bs, p);
    loader.skip_to_end();
    return p;
  }
  // one epoch of training on the device-resident dataset; returns images trained
  long train_epoch_resident(VTCNN &m, Pass &p) {
    const size_t total = train_loader.size(), bs = train_loader.batch(), nb = train_loader.total_batches();
    DeviceCounters dc((int)nb, num_classes, 0);
    const DataLoader::Resident &r = train_loader.resident(m.stream());
    for (size_t b = 0; b < nb; ++b) {
      const size_t first = b * bs, n = std::min(bs, total - first);
      m.train_batch_resident(r.images_u8, r.labels, r.indices + first, (int)n, (float)batch_size, (int)b, dc.loss_sums, dc.correct);
    }
    std::vector<float> ls(nb);
    std::vector<int> ok(nb);
    vt_host::hip_check(hipMemcpyAsync(ls.data(), dc.loss_sums, sizeof(float) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    vt_host::hip_check(hipMemcpyAsync(ok.data(), dc.correct, sizeof(int) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    vt_host::hip_check(hipStreamSynchronize(m.stream()), "sync");
    fold_batches(ls, ok, total, bs, p);
    train_loader.skip_to_end();
    return (long)total;
  }

  // ---- several loader batches per step on ONE GPU (environment VTCNN_VIRTUAL_RANKS = V; no counterpart in the
  // reference): round t trains loader batches t*V .. t*V+V-1 as V independent mini-batches in one launch set, their
  // gradients summed, one update_weights(V * batch_size): the step of V data-parallel ranks, at about the cost of one
  // batch-of-32 step (which is bound by its chain of dependent launches, not by arithmetic).  The full batches of a last,
  // shorter round are trained the same way; a short last batch is trained alone, as the reference trains it.
  int virtual_ranks(VTCNN &m) const {
    const char *want = std::getenv("VTCNN_VIRTUAL_RANKS");
    const int V = want && *want ? std::atoi(want) : 0;
    if (V < 2) return 1;
    const size_t bs = train_loader.batch();
    if (bs % 4 != 0 || m.dims().input_dim != 28) throw std::runtime_error("VTCNN_VIRTUAL_RANKS needs a batch size that is a multiple of 4 and 28x28 inputs");
    if ((size_t)m.max_batch() < bs * V) throw std::runtime_error("VTCNN_VIRTUAL_RANKS: the model's max_batch is smaller than V * batch_size");
    return V;
  }
  long train_epoch_grouped(VTCNN &m, int V, Pass &p) {
    const size_t total = train_loader.size(), bs = train_loader.batch(), nb = train_loader.total_batches();
    DeviceCounters dc((int)nb, num_classes, 0);
    const DataLoader::Resident &r = train_loader.resident(m.stream());
    const size_t full = total / bs;  // batches of exactly bs images
    for (size_t b = 0; b < full; b += V) {
      const size_t k = std::min<size_t>(V, full - b);
      if (k >= 2)
        m.train_group_resident(r.images_u8, r.labels, r.indices + b * bs, (int)(k * bs), (int)bs, (float)batch_size, (int)b, dc.loss_sums,
                               dc.correct);
      else
        m.train_batch_resident(r.images_u8, r.labels, r.indices + b * bs, (int)bs, (float)batch_size, (int)b, dc.loss_sums, dc.correct);
    }
    if (full < nb)  // the short last batch, alone, with the constant divisor (utils/trainer.hpp:53)
      m.train_batch_resident(r.images_u8, r.labels, r.indices + full * bs, (int)(total - full * bs), (float)batch_size, (int)full,
                             dc.loss_sums, dc.correct);
    std::vector<float> ls(nb);
    std::vector<int> ok(nb);
    vt_host::hip_check(hipMemcpyAsync(ls.data(), dc.loss_sums, sizeof(float) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    vt_host::hip_check(hipMemcpyAsync(ok.data(), dc.correct, sizeof(int) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
    vt_host::hip_check(hipStreamSynchronize(m.stream()), "sync");
    fold_batches(ls, ok, total, bs, p);
    train_loader.skip_to_end();
    return (long)total;
  }

  // ---- data parallel over the GPUs of the node from this one process (environment VTCNN_GPUS = G; unset: the
  // single-device loops above.  No counterpart in the reference).  Rank r owns a replica of the model and of the
  // resident training set on a device of its own and a host thread that issues its work; round t trains loader
  // batches t*G .. t*G+G-1, one per rank, the gradient arenas are summed across the ranks and every rank applies
  // update_weights(G * batch_size): one SGD step on G batches per round.  Every rank's BatchNorm layers see the
  // statistics of its own batch only; a last round with fewer than G batches is not trained; validation and the
  // test sweep run on rank 0, whose model is the caller's.  With G = 1 the sum is the identity and the run is the
  // single-device one, byte for byte.
  struct DataParallel {
    vt_dp *dp = nullptr;
    std::vector<std::unique_ptr<VTCNN>> replicas;  // ranks 1 .. G-1
    std::vector<VTCNN *> models;                   // rank -> model (rank 0: the caller's)
    std::vector<unsigned char *> images;           // rank -> resident u8 image matrix on that rank's device
    std::vector<int *> labels, indices;
    int trainable = 0;
    // bucket boundaries of the overlapped exchange inside the gradient arena: [0, att_first) BatchNorm2D + convolution,
    // [att_first, att_end) the attention block, [att_end, fc_first) projector + feed-forward + norms, [fc_first, trainable) fc
    int att_first = 0, att_end = 0, fc_first = 0;
    bool overlap = false;
    // per evaluated loader and rank: a copy of its resident dataset on that rank's device (rank 0: the loader's own)
    struct EvalCopy {
      std::vector<unsigned char *> images;
      std::vector<int *> labels, indices;
    };
    std::map<const DataLoader *, EvalCopy> eval_copies;
    ~DataParallel() {
      for (auto &kv : eval_copies)
        for (size_t r = 1; r < kv.second.images.size(); ++r) {
          (void)hipFree(kv.second.images[r]);
          (void)hipFree(kv.second.labels[r]);
          (void)hipFree(kv.second.indices[r]);
        }
      for (size_t r = 1; r < images.size(); ++r) {  // rank 0's buffers belong to the loader
        (void)hipFree(images[r]);
        (void)hipFree(labels[r]);
        (void)hipFree(indices[r]);
      }
      replicas.clear();
      if (dp) vt_dp_destroy(dp);
    }
  };
  std::unique_ptr<DataParallel> dp_;
  DataParallel *data_parallel(VTCNN &m) {
    const char *want = std::getenv("VTCNN_GPUS");
    if (!want || !*want) return nullptr;
    if (dp_) return dp_.get();
    int have = 0;
    vt_host::hip_check(hipGetDeviceCount(&have), "hipGetDeviceCount");
    const int G = std::atoi(want);
    if (G < 1 || G > have) throw std::runtime_error("VTCNN_GPUS must be between 1 and the number of visible GPUs");
    // one rank has nothing to exchange: the single-device loop, without opening the collective library
    // (VTCNN_DP_FORCE=1 keeps the data-parallel driver even then: a test aid)
    const char *force = std::getenv("VTCNN_DP_FORCE");
    if (G == 1 && !(force && *force && *force != '0')) return nullptr;
    std::unique_ptr<DataParallel> g(new DataParallel);
    std::vector<int> devices{m.device()};
    for (int d = 0; d < have && (int)devices.size() < G; ++d)
      if (d != m.device()) devices.push_back(d);
    const int rc = vt_dp_init(G, devices.data(), &g->dp);
    if (rc != 0) throw std::runtime_error(std::string("vt_dp_init: ") + vt_error_string(rc));
    vt_layout lo;
    vt_model_layout(m.handle(), &lo);
    g->trainable = lo.n_trainable;
    {
      int count = 0;
      vt_model_attention_range(m.handle(), &g->att_first, &count);
      g->att_end = g->att_first + count;
      g->fc_first = lo.t[vt_find(&lo, "fc.W")].offset;
      const char *ov = std::getenv("VTCNN_DP_OVERLAP");
      g->overlap = ov && *ov && *ov != '0';
    }
    g->models.push_back(&m);
    for (int r = 1; r < G; ++r) {
      g->replicas.push_back(m.replica_on(devices[r]));
      g->models.push_back(g->replicas.back().get());
    }
    dp_ = std::move(g);
    return dp_.get();
  }
  // every rank's copy of the resident training set (rank 0: the loader's own), made before the first training epoch
  void replicate_training_set(DataParallel &g) {
    if (!g.images.empty()) return;
    VTCNN &m = *g.models[0];
    const DataLoader::Resident &r0 = train_loader.resident(m.stream());
    const size_t count = train_loader.size(), bytes = count * (size_t)r0.pixels;
    g.images.push_back(r0.images_u8);
    g.labels.push_back(r0.labels);
    g.indices.push_back(r0.indices);
    for (size_t r = 1; r < g.models.size(); ++r) {
      const int dev = g.models[r]->device();
      vt_host::DeviceScope on_device(dev);
      unsigned char *img = nullptr;
      int *lab = nullptr, *idx = nullptr;
      vt_host::hip_check(hipMalloc((void **)&img, bytes), "hipMalloc dataset");
      vt_host::hip_check(hipMalloc((void **)&lab, count * sizeof(int)), "hipMalloc labels");
      vt_host::hip_check(hipMalloc((void **)&idx, count * sizeof(int)), "hipMalloc indices");
      g.images.push_back(img);
      g.labels.push_back(lab);
      g.indices.push_back(idx);
      vt_host::hip_check(hipMemcpyPeer(img, dev, r0.images_u8, m.device(), bytes), "dataset to peer");
      vt_host::hip_check(hipMemcpyPeer(lab, dev, r0.labels, m.device(), count * sizeof(int)), "labels to peer");
    }
  }
  long train_epoch_data_parallel(DataParallel &g, Pass &p) {
    const size_t total = train_loader.size(), bs = train_loader.batch(), nb = train_loader.total_batches();
    const size_t G = g.models.size(), rounds = nb / G;
    if (rounds == 0) throw std::runtime_error("VTCNN_GPUS: the training set has fewer batches than GPUs");
    replicate_training_set(g);
    (void)train_loader.resident(g.models[0]->stream());  // this epoch's order on rank 0's device
    std::vector<std::vector<float>> ls(G, std::vector<float>(rounds));
    std::vector<std::vector<int>> ok(G, std::vector<int>(rounds));
    auto rank_epoch = [&](size_t r) {
      try {
        VTCNN &m = *g.models[r];
        vt_host::DeviceScope on_device(m.device());
        DeviceCounters dc((int)std::max<size_t>(rounds, 1), num_classes, 0);
        if (r > 0)
          vt_host::hip_check(hipMemcpyAsync(g.indices[r], train_loader.order().data(), total * sizeof(int), hipMemcpyHostToDevice,
                                            m.stream()), "H2D indices");
        for (size_t t = 0; t < rounds; ++t) {
          const size_t first = (t * G + r) * bs, n = std::min(bs, total - first);
          float *grads = vt_model_grads_ptr(m.handle());
          if (g.overlap) {
            // VTCNN_DP_OVERLAP=1: the exchange in buckets on a stream of its own, in the order in which gradients
            // become final, beside the attention backward (the longest kernels of the step); the update waits for all
            auto bucket = [&](int lo_, int hi_) {
              const int rc = vt_dp_exchange_begin(g.dp, (int)r, grads, lo_, hi_ - lo_, m.stream());
              if (rc != 0) throw std::runtime_error(std::string("vt_dp_exchange_begin: ") + vt_error_string(rc));
            };
            m.gradients_main_resident(g.images[r], g.labels[r], g.indices[r] + first, (int)n, (int)t, dc.loss_sums, dc.correct);
            bucket(g.fc_first, g.trainable);
            bucket(g.att_end, g.fc_first);
            bucket(0, g.att_first);
            m.gradients_attention();
            bucket(g.att_first, g.att_end);
            const int rc = vt_dp_exchange_join(g.dp, (int)r, m.stream());
            if (rc != 0) throw std::runtime_error(std::string("vt_dp_exchange_join: ") + vt_error_string(rc));
          } else {
            m.gradients_resident(g.images[r], g.labels[r], g.indices[r] + first, (int)n, (int)t, dc.loss_sums, dc.correct);
            const int rc = vt_dp_allreduce_rank_async(g.dp, (int)r, grads, 0, g.trainable, m.stream());
            if (rc != 0) throw std::runtime_error(std::string("vt_dp_allreduce_rank_async: ") + vt_error_string(rc));
          }
          m.update_weights((float)(G * (size_t)batch_size));
        }
        vt_host::hip_check(hipMemcpyAsync(ls[r].data(), dc.loss_sums, sizeof(float) * rounds, hipMemcpyDeviceToHost, m.stream()), "D2H");
        vt_host::hip_check(hipMemcpyAsync(ok[r].data(), dc.correct, sizeof(int) * rounds, hipMemcpyDeviceToHost, m.stream()), "D2H");
        vt_host::hip_check(hipStreamSynchronize(m.stream()), "sync");
      } catch (const std::exception &e) {
        // the other ranks are (or will be) blocked inside the exchange waiting for this one: joining them would hang,
        // so the run ends here, loudly
        std::cerr << "❌ data-parallel rank " << r << " failed: " << e.what() << std::endl;
        std::_Exit(3);
      } catch (...) {
        std::cerr << "❌ data-parallel rank " << r << " failed" << std::endl;
        std::_Exit(3);
      }
    };
    std::vector<std::thread> workers;
    for (size_t r = 1; r < G; ++r) workers.emplace_back(rank_epoch, r);
    rank_epoch(0);
    for (std::thread &w : workers) w.join();
    long images = 0;
    for (size_t t = 0; t < rounds; ++t)  // per-batch means folded in loader order, as the single-device loop does
      for (size_t r = 0; r < G; ++r) {
        const float n = (float)std::min(bs, total - (t * G + r) * bs);
        p.loss += ls[r][t] / n;
        p.acc += static_cast<float>(ok[r][t]) / n;
        ++p.batches;
        images += (long)n;
      }
    train_loader.skip_to_end();
    return images;
  }

  // Evaluation sharded over the replicas ("replicas only": no collective).  The chunks of consecutive loader batches
  // that sweep_resident would run one after another are dealt round-robin to the ranks, each on its own device with a
  // copy of rank 0's weights and BatchNorm2D running statistics; every loader batch keeps its own BatchNorm1D
  // statistics, so the per-batch losses, hit counts and predicted labels are what the single-device sweep computes.
  // The ranks' per-batch numbers and label arrays are merged on the host by owner.
  Pass sweep_data_parallel(DataParallel &g, DataLoader &loader, bool keep_labels) {
    Pass p;
    const size_t total = loader.size(), bs = loader.batch(), nb = loader.total_batches(), G = g.models.size();
    VTCNN &m0 = *g.models[0];
    const DataLoader::Resident &r0 = loader.resident(m0.stream());
    DataParallel::EvalCopy &ec = g.eval_copies[&loader];
    if (ec.images.empty()) {
      const size_t bytes = total * (size_t)r0.pixels;
      ec.images.assign(G, nullptr);
      ec.labels.assign(G, nullptr);
      ec.indices.assign(G, nullptr);
      ec.images[0] = r0.images_u8;
      ec.labels[0] = r0.labels;
      for (size_t r = 1; r < G; ++r) {
        vt_host::DeviceScope on_device(g.models[r]->device());
        vt_host::hip_check(hipMalloc((void **)&ec.images[r], bytes), "hipMalloc dataset");
        vt_host::hip_check(hipMalloc((void **)&ec.labels[r], total * sizeof(int)), "hipMalloc labels");
        vt_host::hip_check(hipMalloc((void **)&ec.indices[r], total * sizeof(int)), "hipMalloc indices");
        vt_host::hip_check(hipMemcpyPeer(ec.images[r], g.models[r]->device(), r0.images_u8, m0.device(), bytes), "dataset to peer");
        vt_host::hip_check(hipMemcpyPeer(ec.labels[r], g.models[r]->device(), r0.labels, m0.device(), total * sizeof(int)), "labels to peer");
      }
    }
    ec.indices[0] = r0.indices;
    for (size_t r = 1; r < G; ++r) g.models[r]->copy_weights_from(m0);  // also rank 0's running statistics
    const size_t per_chunk = (bs % 4 == 0) ? std::max<size_t>(1, (size_t)m0.max_batch() / bs) : 1;
    std::vector<std::vector<float>> ls(G, std::vector<float>(nb));
    std::vector<std::vector<int>> ok(G, std::vector<int>(nb)), am(G);
    auto rank_sweep = [&](size_t r) {
      try {
        VTCNN &m = *g.models[r];
        vt_host::DeviceScope on_device(m.device());
        DeviceCounters dc((int)nb, num_classes, (int)total);
        if (r > 0)
          vt_host::hip_check(hipMemcpyAsync(ec.indices[r], loader.order().data(), total * sizeof(int), hipMemcpyHostToDevice, m.stream()),
                             "H2D indices");
        m.set_training(false);
        size_t chunk = 0;
        for (size_t b0 = 0; b0 < nb; b0 += per_chunk, ++chunk) {
          if (chunk % G != r) continue;
          const size_t first = b0 * bs, n = std::min(per_chunk * bs, total - first);
          m.eval_chunk_resident(ec.images[r], ec.labels[r], ec.indices[r] + first, (int)n, (int)bs, (int)b0, dc.loss_sums, dc.correct,
                                dc.confusion, dc.argmax + first);
        }
        m.set_training(true);
        vt_host::hip_check(hipMemcpyAsync(ls[r].data(), dc.loss_sums, sizeof(float) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
        vt_host::hip_check(hipMemcpyAsync(ok[r].data(), dc.correct, sizeof(int) * nb, hipMemcpyDeviceToHost, m.stream()), "D2H");
        if (keep_labels) {
          am[r].resize(total);
          vt_host::hip_check(hipMemcpyAsync(am[r].data(), dc.argmax, sizeof(int) * total, hipMemcpyDeviceToHost, m.stream()), "D2H");
        }
        vt_host::hip_check(hipStreamSynchronize(m.stream()), "sync");
      } catch (const std::exception &e) {
        std::cerr << "❌ evaluation on rank " << r << " failed: " << e.what() << std::endl;
        std::_Exit(3);
      }
    };
    std::vector<std::thread> workers;
    for (size_t r = 1; r < G; ++r) workers.emplace_back(rank_sweep, r);
    rank_sweep(0);
    for (std::thread &w : workers) w.join();
    std::vector<float> lsm(nb);
    std::vector<int> okm(nb);
    if (keep_labels) {
      p.pred.resize(total);
      p.truth.resize(total);
      for (size_t i = 0; i < total; ++i) p.truth[i] = loader.all_labels()[loader.order()[i]];
    }
    for (size_t b = 0; b < nb; ++b) {
      const size_t owner = (b / per_chunk) % G;
      lsm[b] = ls[owner][b];
      okm[b] = ok[owner][b];
      if (keep_labels)
        for (size_t i = b * bs; i < std::min(total, (b + 1) * bs); ++i) p.pred[i] = am[owner][i];
    }
    fold_batches(lsm, okm, total, bs, p);
    loader.skip_to_end();
    return p;
  }

  // forward-only sweep over a loader in eval mode
  Pass sweep(DataLoader &loader, bool keep_labels) {
    if (VTCNN *m = device_model(loader)) {
      if (DataParallel *g = data_parallel(*m)) return sweep_data_parallel(*g, loader, keep_labels);
      return sweep_resident(*m, loader, keep_labels);
    }
    Pass p;
    model.set_training(false);
    while (loader.has_next()) {
      auto batch = loader.next_batch();
      const Tensor &X = batch.first, &y = batch.second;
      const Tensor onehot = Tensor::one_hot(y, num_classes);
      const Tensor preds = model.forward(X);
      p.loss += Tensor::compute_loss(preds, onehot);
      p.acc += Tensor::compute_accuracy(preds, y, num_classes);
      ++p.batches;
      if (keep_labels) {
        const std::vector<int> yp = Tensor::argmax_batch(preds), yt = y.to_vector();
        p.pred.insert(p.pred.end(), yp.begin(), yp.end());
        p.truth.insert(p.truth.end(), yt.begin(), yt.end());
      }
    }
    model.set_training(true);
    return p;
  }

 public:
  Trainer(Model &model_, DataLoader &train_loader_, DataLoader &val_loader_, int num_classes_, int batch_size_,
          DataLoader *test_loader_ = nullptr)
      : model(model_), train_loader(train_loader_), val_loader(val_loader_), test_loader(test_loader_),
        num_classes(num_classes_), batch_size(batch_size_) {}

  void train(int epochs, int /*log_every*/, std::string /*device*/) {
    for (int epoch = 0; epoch < epochs; ++epoch) {
      const auto t0 = std::chrono::high_resolution_clock::now();
      std::cout << "Epoch " << (epoch + 1) << "/" << epochs << "\n";
      train_loader.reset();
      float epoch_loss = 0.0f, epoch_acc = 0.0f;
      int batches = 0;
      long images = 0;
      if (VTCNN *m = device_model(train_loader)) {
        Pass p;
        DataParallel *g = data_parallel(*m);
        const int V = g ? 1 : virtual_ranks(*m);
        images = g ? train_epoch_data_parallel(*g, p) : (V > 1 ? train_epoch_grouped(*m, V, p) : train_epoch_resident(*m, p));
        epoch_loss = p.loss;
        epoch_acc = p.acc;
        batches = p.batches;
      }
      while (train_loader.has_next()) {
        auto batch = train_loader.next_batch();
        const Tensor &X = batch.first, &y = batch.second;
        const Tensor onehot = Tensor::one_hot(y, num_classes);
        const Tensor preds = model.forward(X);
        model.backward(onehot);
        model.update_weights(batch_size);  // the constant loader batch size, even on a short last batch
        model.zero_grad();
        epoch_loss += Tensor::compute_loss(preds, onehot);
        epoch_acc += Tensor::compute_accuracy(preds, y, num_classes);
        ++batches;
        images += X.shape[0];
      }
      const std::chrono::duration<double> dt = std::chrono::high_resolution_clock::now() - t0;
      std::cout << std::fixed << std::setprecision(4);
      std::cout << "Train Loss: " << (epoch_loss / batches) << " | Accuracy: " << (epoch_acc / batches) * 100.0f << "%"
                << " | Time: " << dt.count() << " s | " << (double)images / dt.count() << " images/sec\n";
      evaluate();
    }
    if (test_loader) {
      std::cout << "\n=== Evaluación final sobre conjunto de test ===\n";
      evaluate_test();
    }
  }

  void evaluate() {
    val_loader.reset();
    const Pass p = sweep(val_loader, false);
    std::cout << "Val Loss: " << (p.loss / p.batches) << " | Accuracy: " << (p.acc / p.batches) * 100.0f << "%\n";
  }

  // returns the predicted labels in loader order (the reference returns nothing; the printout is the same)
  std::vector<int> evaluate_test() {
    if (!test_loader) {
      std::cout << "[WARN] No se proporcionó test_loader\n";
      return {};
    }
    test_loader->reset();
    const Pass p = sweep(*test_loader, true);
    std::cout << "[TEST] Final Evaluation — Loss: " << (p.loss / p.batches) << " | Accuracy: " << (p.acc / p.batches) * 100.0f
              << "%\n";
    const Tensor::ClassStats s = Tensor::class_stats(p.truth, p.pred, num_classes);
    const auto cm = Tensor::confusion_matrix(p.truth, p.pred, num_classes);
    std::cout << " - F1 Score (macro):    " << s.macro_f1 << "\n";
    std::cout << " - F1 Score (weighted): " << s.weighted_f1 << "\n";
    std::cout << " - Precision (macro):   " << s.macro_precision << "\n";
    std::cout << " - Recall (macro):      " << s.macro_recall << "\n";
    std::cout << "\n📊 Per-Class Metrics:\n";
    std::cout << std::setw(8) << "Class" << std::setw(12) << "Precision" << std::setw(12) << "Recall" << std::setw(12)
              << "F1 Score" << "\n"
              << std::string(44, '-') << "\n";
    for (int k = 0; k < num_classes; ++k)
      std::cout << std::setw(8) << k << std::fixed << std::setprecision(4) << std::setw(12) << s.precision[k] << std::setw(12)
                << s.recall[k] << std::setw(12) << s.f1[k] << "\n";
    std::cout << "\n🧩 Matriz de Confusión:\n    ";
    for (int j = 0; j < num_classes; ++j) std::cout << std::setw(4) << j;
    std::cout << "\n";
    for (int i = 0; i < num_classes; ++i) {
      std::cout << std::setw(2) << i << ": ";
      for (int j = 0; j < num_classes; ++j) std::cout << std::setw(4) << cm[i][j];
      std::cout << "\n";
    }
    return p.pred;
  }

  // single 28x28 image, eval mode, first maximum wins
  int predict(const std::vector<float> &image_data) {
    if (image_data.size() != 28 * 28) throw std::runtime_error("La imagen debe ser de tamaño 28x28");
    const Tensor input({1, 1, 28, 28}, image_data);
    model.set_training(false);
    const Tensor out = model.forward(input);
    (void)out.printsummary();  // the reference builds this string and drops it (utils/trainer.hpp:208): nothing is printed
    int best = 0;
    for (size_t i = 1; i < out.data.size(); ++i)
      if (out.data[i] > out.data[best]) best = (int)i;
    return best;
  }
};
