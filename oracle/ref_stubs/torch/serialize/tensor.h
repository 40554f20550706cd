// Stub used ONLY when compiling the reference's *_kernel.cu files into oracle/_ref/
// (oracle/Makefile target `ref`).  The reference kernel headers include torch headers
// solely to declare at::Tensor wrapper prototypes that the kernel translation units
// never define or call; a forward declaration is all they need.
#pragma once
namespace at { class Tensor; }
