// load_image_as_vector with the reference's signature, texts and pixel values
// (utils/load_image_stb.hpp:9-26: stbi_load(..., 1) then pixel / 255.0f).  The decode is the C ABI's
// vt_image_read_gray, which reproduces that decoder's grey byte for every PNG colour type and for baseline
// JPEG files; what it does not decode is rejected with the reference's "could not load" text.
#pragma once
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/vtcnn_hip.h"

inline std::vector<float> load_image_as_vector(const std::string &path, int target_w = 28, int target_h = 28) {
  int width = 0, height = 0;
  if (vt_image_read_gray(path.c_str(), nullptr, 0, &width, &height) != 0)
    throw std::runtime_error("No se pudo cargar la imagen: " + path);
  if (width != target_w || height != target_h) throw std::runtime_error("La imagen debe ser de tamaño 28x28");
  std::vector<unsigned char> raw((size_t)width * height);
  if (vt_image_read_gray(path.c_str(), raw.data(), (int)raw.size(), &width, &height) != 0)
    throw std::runtime_error("No se pudo cargar la imagen: " + path);
  std::vector<float> image((size_t)target_w * target_h);
  for (size_t i = 0; i < image.size(); ++i) image[i] = static_cast<float>(raw[i]) / 255.0f;
  return image;
}
