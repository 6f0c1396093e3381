// Command line of the reference (main.cpp:219-277):
//   main <mnist|bloodmnist|fashionmnist> <train|predict> [epochs]
// with the same literal hyper-parameters (main.cpp:66-78), the same 80/20 train/validation split
// and the same console lines (the web frontend greps "Imagen predicción:").  Runs on HIP device 0.
// Environment overrides that do not change the positional CLI: VTCNN_BATCH_SIZE (default 32),
// VTCNN_NUM_CLASSES (default 10), VTCNN_DATA_ROOT (default "."), VTCNN_LABELS_OUT (predict mode:
// file that receives the predicted test labels, one per line, in loader order), VTCNN_EVAL_CHUNK (evaluate this
// many images per launch set as consecutive loader batches with their own BatchNorm1D statistics; default:
// one loader batch at a time, which is bit-identical to the host loop, while chunks differ from it in the
// last bits of the running statistics), VTCNN_HOST_LOOP=1 (the reference's per-batch host loop instead of the
// device-resident one), VTCNN_GPUS (train data-parallel over that many GPUs of the node: one replica and one
// loader batch per GPU and round, gradients summed over RCCL, see utils/trainer.hpp), VTCNN_VIRTUAL_RANKS (train that
// many consecutive loader batches per step on one GPU as independent mini-batches with one summed update).
#include <algorithm>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <string>

#include "model/vit_cnn.hpp"
#include "utils/load_dataset.hpp"
#include "utils/trainer.hpp"

std::mt19937 Tensor::global_gen(42);

namespace {
struct DatasetInfo {
  const char *name, *dir, *model_file;
};
const DatasetInfo kDatasets[] = {{"mnist", "mnist_data/", "model_mnist.bin"},
                                 {"bloodmnist", "blood_data/", "model_bloodmnist.bin"},
                                 {"fashionmnist", "fashion_data/", "model_fashionmnist.bin"}};

int env_int(const char *name, int dflt) {
  const char *v = std::getenv(name);
  return v && *v ? std::atoi(v) : dflt;
}

void run_pipeline(const DatasetInfo &ds, bool train_mode, const std::string &device, int epochs) {
  const int image_channels = 1, input_dim = 28, vt_layers_num = 1, tokens = 16, token_channels = 32, vt_channels = 16;
  const int transformer_enc_layers = 2, transformer_heads = 2, transformer_fc_dim = 64;
  const float transformer_dropout = 0.1f;
  const int num_classes = env_int("VTCNN_NUM_CLASSES", 10);
  const int batch_size = env_int("VTCNN_BATCH_SIZE", 32);
  const int max_samples = 60000;
  const char *root_env = std::getenv("VTCNN_DATA_ROOT");
  const std::string root = root_env && *root_env ? std::string(root_env) + "/" : std::string();
  const std::string path = root + ds.dir, model_file = root + ds.model_file;

  Dataset full_train = load_dataset(path + "train-images-idx3-ubyte", path + "train-labels-idx1-ubyte", max_samples);
  Dataset test_set = load_dataset(path + "t10k-images-idx3-ubyte", path + "t10k-labels-idx1-ubyte", 10000);
  const size_t total = full_train.images.size(), train_count = static_cast<size_t>(total * 0.8);
  Dataset train_dataset, val_dataset;
  train_dataset.images.assign(full_train.images.begin(), full_train.images.begin() + train_count);
  train_dataset.labels.assign(full_train.labels.begin(), full_train.labels.begin() + train_count);
  val_dataset.images.assign(full_train.images.begin() + train_count, full_train.images.end());
  val_dataset.labels.assign(full_train.labels.begin() + train_count, full_train.labels.end());
  std::cout << "📊 Datos cargados:\n";
  std::cout << "  - Train: " << train_dataset.images.size() << " muestras\n";
  std::cout << "  - Validation: " << val_dataset.images.size() << " muestras\n";
  std::cout << "  - Test: " << test_set.images.size() << " muestras\n";

  DataLoader train_loader(train_dataset.images, train_dataset.labels, batch_size);
  DataLoader val_loader(val_dataset.images, val_dataset.labels, batch_size);
  DataLoader test_loader(test_set.images, test_set.labels, batch_size);

  VTCNN model(image_channels, input_dim, vt_layers_num, tokens, token_channels, vt_channels, transformer_enc_layers,
              transformer_heads, transformer_fc_dim, transformer_dropout, num_classes,
              // workspace: evaluation runs VTCNN_EVAL_CHUNK images per launch set, a grouped training step
              // VTCNN_VIRTUAL_RANKS loader batches
              std::max({batch_size, env_int("VTCNN_EVAL_CHUNK", 0), batch_size * std::max(1, env_int("VTCNN_VIRTUAL_RANKS", 1))}));
  Trainer trainer(model, train_loader, val_loader, num_classes, batch_size, &test_loader);
  if (train_mode) {
    std::cout << "🚀 Entrenando modelo por " << epochs << " épocas...\n";
    trainer.train(epochs, 100, device);
    model.save(model_file);
    std::cout << "✅ Modelo guardado como: " << model_file << "\n";
    return;
  }
  try {
    model.load(model_file);
    model.set_training(false);
    std::cout << "✅ Modelo cargado desde: " << model_file << "\n";
  } catch (const std::exception &e) {
    std::cerr << "❌ Error al cargar modelo: " << e.what() << "\n";
    return;
  }
  std::cout << "\n=== Evaluación del modelo cargado ===\n";
  const std::vector<int> pred = trainer.evaluate_test();
  if (const char *out = std::getenv("VTCNN_LABELS_OUT")) {
    std::ofstream f(out);
    for (int p : pred) f << p << "\n";
  }
  std::cout << "\n🔍 Cargando imágenes personalizadas desde 'custom_images/'...\n";
  Dataset custom = load_custom_images_from_folder(root + "custom_images/");
  for (size_t i = 0; i < custom.images.size(); ++i) {
    const int k = trainer.predict(custom.images[i]);
    std::cout << "Imagen predicción: " << k << "\n";
  }
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3) {
    std::cerr << "Uso: ./main <mnist|bloodmnist|fashionmnist> <train|predict> [epocas]\n";
    return 1;
  }
  std::string dataset_str = argv[1], mode_str = argv[2];
  std::transform(dataset_str.begin(), dataset_str.end(), dataset_str.begin(), ::tolower);
  std::transform(mode_str.begin(), mode_str.end(), mode_str.begin(), ::tolower);
  const DatasetInfo *ds = nullptr;
  for (const DatasetInfo &d : kDatasets)
    if (dataset_str == d.name) ds = &d;
  if (!ds) {
    std::cerr << "❌ Dataset inválido: " << dataset_str << "\n";
    return 1;
  }
  if (mode_str != "train" && mode_str != "predict") {
    std::cerr << "❌ Modo inválido (usa 'train' o 'predict'): " << mode_str << "\n";
    return 1;
  }
  const bool train_mode = mode_str == "train";
  int epochs = 5;
  if (train_mode && argc >= 4) {
    try {
      epochs = std::stoi(argv[3]);
      if (epochs <= 0) {
        std::cerr << "❌ Número de épocas inválido.\n";
        return 1;
      }
    } catch (const std::exception &e) {
      std::cerr << "❌ Error al parsear épocas: " << e.what() << "\n";
      return 1;
    }
  }
  try {
    run_pipeline(*ds, train_mode, "hip", epochs);
  } catch (const std::exception &e) {
    std::cerr << "❌ " << e.what() << "\n";
    return 2;
  }
  return 0;
}
