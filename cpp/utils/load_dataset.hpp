// Dataset loading and batching with the reference's semantics (utils/load_dataset.hpp:12-165):
// IDX files (big-endian headers, pixel u8/255.0f), DataLoader that shuffles its index vector in
// the constructor and again at every reset() from one mt19937(seed), short last batch.
// The bit-exact pieces (IDX decode, shuffle) are the C ABI's host entry points.
#pragma once
#include <algorithm>
#include <cstdio>
#include <filesystem>
#include <iostream>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../include/vtcnn_hip.h"
#include "load_image_stb.hpp"
#include "tensor.hpp"

struct Dataset {
  std::vector<std::vector<float>> images;
  std::vector<int> labels;
};

inline std::vector<std::vector<float>> loadImages2D(const std::string &filename, int max_images = 9999999) {
  int n = 0, r = 0, c = 0;
  int rc = vt_idx_read_images(filename.c_str(), max_images, nullptr, &n, &r, &c);
  if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo de imágenes");
  if (rc != 0) throw std::runtime_error("Archivo de imágenes IDX inválido: " + filename + " (" + vt_error_string(rc) + ")");
  std::vector<float> flat((size_t)n * r * c);
  rc = vt_idx_read_images(filename.c_str(), max_images, flat.data(), &n, &r, &c);
  if (rc != 0) throw std::runtime_error("No se pudo leer el archivo de imágenes: " + filename + " (" + vt_error_string(rc) + ")");
  std::vector<std::vector<float>> out(n);
  for (int i = 0; i < n; ++i) out[i].assign(flat.begin() + (size_t)i * r * c, flat.begin() + (size_t)(i + 1) * r * c);
  return out;
}

inline std::vector<int> loadLabels(const std::string &filename, int max_labels = 9999999) {
  int n = 0;
  int rc = vt_idx_read_labels(filename.c_str(), max_labels, nullptr, &n);
  if (rc == VT_EIO) throw std::runtime_error("No se pudo abrir el archivo de etiquetas");
  if (rc != 0) throw std::runtime_error("Archivo de etiquetas IDX inválido: " + filename + " (" + vt_error_string(rc) + ")");
  std::vector<int> out(n);
  rc = vt_idx_read_labels(filename.c_str(), max_labels, out.data(), &n);
  if (rc != 0) throw std::runtime_error("No se pudo leer el archivo de etiquetas: " + filename + " (" + vt_error_string(rc) + ")");
  return out;
}

inline Dataset load_dataset(const std::string &image_path, const std::string &label_path, int max_samples = 9999999) {
  Dataset d{loadImages2D(image_path, max_samples), loadLabels(label_path, max_samples)};
  if (d.images.size() != d.labels.size()) throw std::runtime_error("El número de imágenes y etiquetas no coincide");
  return d;
}

class DataLoader {
  const std::vector<std::vector<float>> &images;
  const std::vector<int> &labels;
  size_t batch_size, index = 0;
  std::vector<int> indices;
  unsigned seed;
  int resets = 0;
  int C, H, W;

  void reshuffle() {
    indices.resize(images.size());
    if (vt_loader_indices((int)images.size(), seed, resets, indices.data()) != 0)
      throw std::runtime_error("DataLoader: vt_loader_indices failed");
  }

 public:
  DataLoader(const std::vector<std::vector<float>> &imgs, const std::vector<int> &lbls, size_t b_size, int channels = 1,
             int height = 28, int width = 28, int /*num_cls*/ = 10, unsigned int seed_ = 42)
      : images(imgs), labels(lbls), batch_size(b_size), seed(seed_), C(channels), H(height), W(width) {
    if (images.size() != labels.size()) throw std::runtime_error("Mismatch entre imágenes y etiquetas");
    reshuffle();
  }
  std::pair<Tensor, Tensor> next_batch() {
    const size_t end = std::min(index + batch_size, images.size()), n = end - index, per = (size_t)C * H * W;
    std::vector<float> x(n * per), y(n);
    for (size_t i = 0; i < n; ++i) {
      const size_t src = (size_t)indices[index + i];
      std::copy(images[src].begin(), images[src].end(), x.begin() + i * per);
      y[i] = static_cast<float>(labels[src]);
    }
    index = end;
    return {Tensor({(int)n, C, H, W}, std::move(x)), Tensor({(int)n, 1}, std::move(y))};
  }
  bool has_next() const { return index < images.size(); }
  void reset() {
    index = 0;
    ++resets;
    reshuffle();
    indices_uploaded_ = false;
  }
  size_t total_batches() const { return (images.size() + batch_size - 1) / batch_size; }
  const std::vector<int> &order() const { return indices; }
  size_t size() const { return images.size(); }
  size_t batch() const { return batch_size; }
  const std::vector<int> &all_labels() const { return labels; }

  // ---- device residency (no counterpart in the reference, whose next_batch copies every batch on the host,
  // :129-151).  The whole dataset is kept on the device as the IDX bytes it was decoded from (pixel * 255 is an
  // exact integer for data that came from loadImages2D: u8 / 255.0f), with the labels; a batch is then rows
  // indices[first .. first + n) of that matrix, gathered by vt_loader_gather with the same division.  Only
  // the shuffled index vector travels per epoch.  make_resident returns false (and the caller keeps the host
  // loop) if some pixel is not of that form.
  struct Resident {
    unsigned char *images_u8 = nullptr;
    int *labels = nullptr, *indices = nullptr;
    int pixels = 0;
  };
  bool make_resident() {
    if (resident_.images_u8) return true;
    if (resident_failed_ || images.empty()) return false;
    const size_t per = (size_t)C * H * W;
    std::vector<unsigned char> bytes(images.size() * per);
    for (size_t i = 0; i < images.size(); ++i) {
      if (images[i].size() != per) return !(resident_failed_ = true);
      for (size_t p = 0; p < per; ++p) {
        const float v = images[i][p] * 255.0f;
        const int q = (int)(v + 0.5f);
        if (q < 0 || q > 255 || static_cast<float>(q) / 255.0f != images[i][p]) return !(resident_failed_ = true);
        bytes[i * per + p] = (unsigned char)q;
      }
    }
    if (per % 4 != 0) return !(resident_failed_ = true);
    vt_host::hip_check(hipMalloc((void **)&resident_.images_u8, bytes.size()), "hipMalloc dataset");
    vt_host::hip_check(hipMalloc((void **)&resident_.labels, labels.size() * sizeof(int)), "hipMalloc labels");
    vt_host::hip_check(hipMalloc((void **)&resident_.indices, labels.size() * sizeof(int)), "hipMalloc indices");
    vt_host::hip_check(hipMemcpy(resident_.images_u8, bytes.data(), bytes.size(), hipMemcpyHostToDevice), "H2D dataset");
    vt_host::hip_check(hipMemcpy(resident_.labels, labels.data(), labels.size() * sizeof(int), hipMemcpyHostToDevice), "H2D labels");
    resident_.pixels = (int)per;
    indices_uploaded_ = false;
    return true;
  }
  // the resident view with the CURRENT shuffled order on the device (uploaded once per reset)
  const Resident &resident(hipStream_t stream) {
    if (!indices_uploaded_) {
      vt_host::hip_check(hipMemcpyAsync(resident_.indices, indices.data(), indices.size() * sizeof(int), hipMemcpyHostToDevice, stream),
                         "H2D indices");
      vt_host::hip_check(hipStreamSynchronize(stream), "sync");  // `indices` may be reshuffled right after
      indices_uploaded_ = true;
    }
    return resident_;
  }
  void skip_to_end() { index = images.size(); }
  DataLoader(const DataLoader &) = delete;
  DataLoader &operator=(const DataLoader &) = delete;
  ~DataLoader() {
    if (resident_.images_u8) (void)hipFree(resident_.images_u8);
    if (resident_.labels) (void)hipFree(resident_.labels);
    if (resident_.indices) (void)hipFree(resident_.indices);
  }

 private:
  Resident resident_;
  bool resident_failed_ = false, indices_uploaded_ = false;
};

// Custom images for `predict` (utils/load_dataset.hpp:167-209): the .png / .jpg / .jpeg files of the
// folder in alphabetical order of their names, each through load_image_as_vector; a file that does
// not load is reported on stderr and skipped.  Binary PGM (P5, 28x28, maxval 255) is accepted as
// well (an addition: handy for raw dataset pixels).
inline std::vector<float> load_pgm_as_vector(const std::string &path) {
  FILE *f = std::fopen(path.c_str(), "rb");
  if (!f) throw std::runtime_error("No se pudo cargar la imagen: " + path);
  int w = 0, h = 0, mx = 0;
  std::vector<float> img;
  if (std::fscanf(f, "P5 %d %d %d", &w, &h, &mx) == 3 && w == 28 && h == 28 && mx == 255) {
    std::fgetc(f);
    std::vector<unsigned char> raw(784);
    if (std::fread(raw.data(), 1, 784, f) == 784) {
      img.resize(784);
      for (int i = 0; i < 784; ++i) img[i] = static_cast<float>(raw[i]) / 255.0f;
    }
  }
  std::fclose(f);
  if (img.empty()) throw std::runtime_error("La imagen debe ser de tamaño 28x28");
  return img;
}

inline Dataset load_custom_images_from_folder(const std::string &folder_path) {
  namespace fs = std::filesystem;
  Dataset ds;
  std::vector<fs::path> files;
  for (const auto &e : fs::directory_iterator(folder_path)) {  // throws if the folder is missing, like the reference
    if (!e.is_regular_file()) continue;
    const std::string ext = e.path().extension().string();
    if (ext == ".png" || ext == ".jpg" || ext == ".jpeg" || ext == ".pgm") files.push_back(e.path());
  }
  std::sort(files.begin(), files.end(),
            [](const fs::path &a, const fs::path &b) { return a.filename().string() < b.filename().string(); });
  for (const fs::path &p : files) {
    try {
      ds.images.push_back(p.extension() == ".pgm" ? load_pgm_as_vector(p.string()) : load_image_as_vector(p.string()));
      ds.labels.push_back(0);  // dummy label
    } catch (const std::exception &e) {
      std::cerr << "Error al cargar " << p.string() << ": " << e.what() << "\n";
    }
  }
  return ds;
}
