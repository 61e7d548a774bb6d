// pybind_containers.cpp -- the container classes of the reference's Python module (pybind.cpp:63-70): VectorBool,
// VectorFloat, VectorIndividual, VectorUInt, VectorPairObservations, Matrix, UMapFloatToVectorFloat, UMapIntToVectorFloat.
//
// The reference declares these STL types opaque for its whole module, so every vector-valued member is such an object.
// Here the module's own functions keep the list / numpy conversions (pybind_module.cpp); this translation unit -- and only
// this one -- declares the types opaque and binds the classes, so that a caller who constructs, fills and passes these
// objects finds them: they are sequences / mappings, and every function of the module that takes a list takes them too.
#include <pybind11/pybind11.h>
#include <pybind11/stl_bind.h>

#include <unordered_map>
#include <vector>

#include "data.hpp"
#include "hmm.hpp"

using fsmc_host::Individual;
using fsmc_host::PairObservations;

PYBIND11_MAKE_OPAQUE(std::vector<bool>)
PYBIND11_MAKE_OPAQUE(std::vector<float>)
PYBIND11_MAKE_OPAQUE(std::vector<unsigned int>)
PYBIND11_MAKE_OPAQUE(std::vector<std::vector<float>>)
PYBIND11_MAKE_OPAQUE(std::vector<Individual>)
PYBIND11_MAKE_OPAQUE(std::vector<PairObservations>)
PYBIND11_MAKE_OPAQUE(std::unordered_map<float, std::vector<float>>)
PYBIND11_MAKE_OPAQUE(std::unordered_map<int, std::vector<float>>)

namespace py = pybind11;

namespace fsmc_host
{
// called by PYBIND11_MODULE(_pyasmc) after Individual and PairObservations are registered
void bindContainers(py::module_& m)
{
  py::bind_vector<std::vector<bool>>(m, "VectorBool");
  py::bind_vector<std::vector<float>>(m, "VectorFloat");
  py::bind_vector<std::vector<Individual>>(m, "VectorIndividual");
  py::bind_vector<std::vector<unsigned int>>(m, "VectorUInt");
  py::bind_vector<std::vector<PairObservations>>(m, "VectorPairObservations");
  py::bind_vector<std::vector<std::vector<float>>>(m, "Matrix");
  py::bind_map<std::unordered_map<float, std::vector<float>>>(m, "UMapFloatToVectorFloat");
  py::bind_map<std::unordered_map<int, std::vector<float>>>(m, "UMapIntToVectorFloat");
}
} // namespace fsmc_host
