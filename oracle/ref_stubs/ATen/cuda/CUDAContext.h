// Stub, see ../../torch/serialize/tensor.h
#pragma once
namespace at { class Tensor; }
