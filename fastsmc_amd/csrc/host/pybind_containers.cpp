// pybind_containers.cpp -- the container classes of the reference's Python module (pybind.cpp:63-70): VectorBool,
// VectorFloat, VectorIndividual, VectorUInt, VectorPairObservations, Matrix, UMapFloatToVectorFloat, UMapIntToVectorFloat.
//
// The reference declares these STL types opaque for its whole module, so every vector-valued member is such an object.
// Here the module's own functions keep the list / numpy conversions of <pybind11/stl.h> (pybind_module.cpp), so the STL
// types themselves must NOT be opaque anywhere in the module: an opaque declaration in one translation unit next to the
// default list caster in another is two definitions of type_caster<std::vector<float>> in one shared object (an ODR
// violation that only worked by the luck of weak-symbol merging).  The reference's names are therefore bound to
// DISTINCT wrapper types -- `struct VectorFloat : std::vector<float>` and so on -- which no caster of <pybind11/stl.h>
// matches: a caller who constructs, fills and passes these objects finds them, they are sequences / mappings, and
// every function of the module that takes a list takes them too (the list casters accept any sequence).
#include <pybind11/pybind11.h>
#include <pybind11/stl_bind.h>

#include <unordered_map>
#include <vector>

#include "data.hpp"
#include "hmm.hpp"

namespace py = pybind11;

namespace fsmc_host
{
namespace containers
{
struct VectorBool : std::vector<bool> { using std::vector<bool>::vector; };
struct VectorFloat : std::vector<float> { using std::vector<float>::vector; };
struct VectorUInt : std::vector<unsigned int> { using std::vector<unsigned int>::vector; };
struct VectorIndividual : std::vector<Individual> { using std::vector<Individual>::vector; };
struct VectorPairObservations : std::vector<PairObservations> { using std::vector<PairObservations>::vector; };
struct Matrix : std::vector<VectorFloat> { using std::vector<VectorFloat>::vector; };
struct UMapFloatToVectorFloat : std::unordered_map<float, VectorFloat> {
  using std::unordered_map<float, VectorFloat>::unordered_map;
};
struct UMapIntToVectorFloat : std::unordered_map<int, VectorFloat> {
  using std::unordered_map<int, VectorFloat>::unordered_map;
};
} // namespace containers

// called by PYBIND11_MODULE(_pyasmc) after Individual and PairObservations are registered
void bindContainers(py::module_& m)
{
  using namespace containers;
  py::bind_vector<VectorBool>(m, "VectorBool");
  py::bind_vector<VectorFloat>(m, "VectorFloat");
  py::bind_vector<VectorIndividual>(m, "VectorIndividual");
  py::bind_vector<VectorUInt>(m, "VectorUInt");
  py::bind_vector<VectorPairObservations>(m, "VectorPairObservations");
  py::bind_vector<Matrix>(m, "Matrix");
  py::bind_map<UMapFloatToVectorFloat>(m, "UMapFloatToVectorFloat");
  py::bind_map<UMapIntToVectorFloat>(m, "UMapIntToVectorFloat");
}
} // namespace fsmc_host
