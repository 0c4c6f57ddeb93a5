// Native model-directory loader (loader.cpp).
#pragma once
#include <string>

#include "model.hpp"

namespace qtts {

// parse_config + every tensor of the directory's safetensors files + finalize, on a Model whose ctx and dtype are set
void load_model_dir(Model& m, const std::string& dir, bool load_tokenizer, bool load_encoder);

}  // namespace qtts
