// pybind_module.cpp -- Python surface of the host library: the names, signatures and field names of the
// reference's pyASMC module (ASMC_SRC/SRC/pybind.cpp:54-252; re-exported as package `asmc`,
// __init__.py:18-33) over the MI355X engine.  Arrays come back as numpy arrays with the reference's
// shapes ((S,K) sums, (K,S) posteriors, (P,S) means / MAPs).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "binary_reader.hpp"
#include "drivers.hpp"
#include "hashing.hpp"
#include "util.hpp"

namespace py = pybind11;
using namespace py::literals;
using namespace fsmc_host;

namespace
{

template <typename T> py::array_t<T> toArray(const std::vector<T>& v, std::vector<py::ssize_t> shape)
{
  py::array_t<T> a(shape);
  if (!v.empty()) {
    std::copy(v.begin(), v.end(), a.mutable_data());
  }
  return a;
}

py::array_t<float> mat2(const std::vector<float>& v, long rows, long cols)
{
  if (v.empty()) {
    return py::array_t<float>(std::vector<py::ssize_t>{0, 0});
  }
  return toArray<float>(v, {rows, cols});
}

py::dict keyedTableToDict(const KeyedTable& t, unsigned states)
{
  py::dict d;
  for (size_t r = 0; r < t.size(); ++r) {
    d[py::float_(t.keys[r])] = std::vector<float>(t.row(static_cast<int>(r), states), t.row(static_cast<int>(r), states) + states);
  }
  return d;
}

std::vector<float> asFloatVec(const py::array_t<float, py::array::c_style | py::array::forcecast>& a)
{
  return std::vector<float>(a.data(), a.data() + a.size());
}

std::vector<std::vector<std::vector<float>>> asCube(const py::array_t<float, py::array::c_style | py::array::forcecast>& a)
{
  if (a.ndim() != 3) {
    throw std::runtime_error("expected a 3-d array [undistinguished][distinguished][states]");
  }
  std::vector<std::vector<std::vector<float>>> out(static_cast<size_t>(a.shape(0)));
  for (py::ssize_t u = 0; u < a.shape(0); ++u) {
    out[u].resize(static_cast<size_t>(a.shape(1)));
    for (py::ssize_t d = 0; d < a.shape(1); ++d) {
      const float* p = a.data(u, d, 0);
      out[u][d].assign(p, p + a.shape(2));
    }
  }
  return out;
}

std::vector<std::vector<float>> asMat(const py::array_t<float, py::array::c_style | py::array::forcecast>& a)
{
  if (a.ndim() != 2) {
    throw std::runtime_error("expected a 2-d array");
  }
  std::vector<std::vector<float>> out(static_cast<size_t>(a.shape(0)));
  for (py::ssize_t r = 0; r < a.shape(0); ++r) {
    out[r].assign(a.data(r, 0), a.data(r, 0) + a.shape(1));
  }
  return out;
}

} // namespace

namespace fsmc_host
{
void bindContainers(py::module_& m); // pybind_containers.cpp: VectorBool ... UMapIntToVectorFloat (pybind.cpp:63-70)
}

PYBIND11_MODULE(_pyasmc, m)
{
  m.doc() = "MI355X-native drop-in for the decode path of PalamaraLab/FastSMC (pyASMC-compatible names)";

  py::enum_<DecodingModeOverall>(m, "DecodingModeOverall", py::arithmetic())
      .value("sequence", DecodingModeOverall::sequence)
      .value("array", DecodingModeOverall::array);
  py::enum_<DecodingMode>(m, "DecodingMode", py::arithmetic())
      .value("sequenceFolded", DecodingMode::sequenceFolded)
      .value("arrayFolded", DecodingMode::arrayFolded)
      .value("sequence", DecodingMode::sequence)
      .value("array", DecodingMode::array);

  py::class_<DecodingReturnValues>(m, "DecodingReturnValues")
      .def_property_readonly("sumOverPairs", [](const DecodingReturnValues& r) { return mat2(r.sumOverPairs, r.sites, r.states); })
      .def_property_readonly("sumOverPairs00", [](const DecodingReturnValues& r) { return mat2(r.sumOverPairs00, r.sites, r.states); })
      .def_property_readonly("sumOverPairs01", [](const DecodingReturnValues& r) { return mat2(r.sumOverPairs01, r.sites, r.states); })
      .def_property_readonly("sumOverPairs11", [](const DecodingReturnValues& r) { return mat2(r.sumOverPairs11, r.sites, r.states); })
      .def_readwrite("sites", &DecodingReturnValues::sites)
      .def_readwrite("states", &DecodingReturnValues::states)
      .def_readwrite("siteWasFlippedDuringFolding", &DecodingReturnValues::siteWasFlippedDuringFolding);

  py::class_<DecodePairsReturnStruct>(m, "DecodePairsReturnStruct")
      .def_readwrite("per_pair_indices", &DecodePairsReturnStruct::perPairIndices)
      .def_property_readonly("per_pair_posteriors",
                             [](const DecodePairsReturnStruct& r) {
                               py::list out;
                               for (const auto& p : r.perPairPosteriors) {
                                 out.append(mat2(p, r.numStates, r.numSites));
                               }
                               return out;
                             })
      .def_property_readonly("sum_of_posteriors", [](const DecodePairsReturnStruct& r) { return mat2(r.sumOfPosteriors, r.numStates, r.numSites); })
      .def_property_readonly("per_pair_posterior_means", [](const DecodePairsReturnStruct& r) { return mat2(r.perPairPosteriorMeans, r.numPairs, r.numSites); })
      .def_property_readonly("min_posterior_means", [](const DecodePairsReturnStruct& r) { return toArray<float>(r.minPosteriorMeans, {static_cast<py::ssize_t>(r.minPosteriorMeans.size())}); })
      .def_property_readonly("argmin_posterior_means", [](const DecodePairsReturnStruct& r) { return toArray<int>(r.argminPosteriorMeans, {static_cast<py::ssize_t>(r.argminPosteriorMeans.size())}); })
      .def_property_readonly("per_pair_MAPs",
                             [](const DecodePairsReturnStruct& r) {
                               if (r.perPairMAPs.empty()) {
                                 return py::array_t<int>(std::vector<py::ssize_t>{0, 0});
                               }
                               return toArray<int>(r.perPairMAPs, {r.numPairs, r.numSites});
                             })
      .def_property_readonly("min_MAPs", [](const DecodePairsReturnStruct& r) { return toArray<int>(r.minMAPs, {static_cast<py::ssize_t>(r.minMAPs.size())}); })
      .def_property_readonly("argmin_MAPs", [](const DecodePairsReturnStruct& r) { return toArray<int>(r.argminMAPs, {static_cast<py::ssize_t>(r.argminMAPs.size())}); });

  py::class_<PairObservations>(m, "PairObservations")
      .def_readwrite("obsBits", &PairObservations::obsBits)
      .def_readwrite("homMinorBits", &PairObservations::homMinorBits)
      .def_readwrite("iHap", &PairObservations::iHap)
      .def_readwrite("jHap", &PairObservations::jHap)
      .def_readwrite("iInd", &PairObservations::iInd)
      .def_readwrite("jInd", &PairObservations::jInd);

  py::class_<DecodingQuantities>(m, "DecodingQuantities")
      .def(py::init<const std::string&>())
      .def_static(
          "from_arrays",
          [](int csfsSamples, py::array_t<float, py::array::c_style | py::array::forcecast> discretization,
             py::array_t<float, py::array::c_style | py::array::forcecast> expectedTimes,
             py::array_t<float, py::array::c_style | py::array::forcecast> initialStateProb,
             py::array_t<float, py::array::c_style | py::array::forcecast> columnRatios,
             py::array_t<float, py::array::c_style | py::array::forcecast> keys,
             py::array_t<float, py::array::c_style | py::array::forcecast> D,
             py::array_t<float, py::array::c_style | py::array::forcecast> B,
             py::array_t<float, py::array::c_style | py::array::forcecast> U,
             py::array_t<float, py::array::c_style | py::array::forcecast> RR,
             py::array_t<float, py::array::c_style | py::array::forcecast> compressedEmission,
             py::array_t<float, py::array::c_style | py::array::forcecast> classicEmission,
             py::array_t<float, py::array::c_style | py::array::forcecast> foldedAscertainedCSFS,
             py::array_t<float, py::array::c_style | py::array::forcecast> ascertainedCSFS,
             py::object CSFS, py::object foldedCSFS, py::object homozygousKeys, py::object homozygousEmissions) {
            DecodingQuantities q;
            q.states = static_cast<unsigned>(expectedTimes.size());
            q.CSFSSamples = csfsSamples;
            q.discretization = asFloatVec(discretization);
            q.expectedTimes = asFloatVec(expectedTimes);
            q.initialStateProb = asFloatVec(initialStateProb);
            q.columnRatios = asFloatVec(columnRatios);
            q.columnRatios.resize(q.states, 0.f);
            const auto k = asFloatVec(keys);
            const auto fill = [&](KeyedTable& t, const py::array_t<float, py::array::c_style | py::array::forcecast>& a) {
              if (a.ndim() != 2 || static_cast<size_t>(a.shape(0)) != k.size() || a.shape(1) != static_cast<py::ssize_t>(q.states)) {
                throw std::runtime_error("transition table must be [keys][states]");
              }
              for (size_t r = 0; r < k.size(); ++r) {
                t.add(k[r], std::vector<float>(a.data(r, 0), a.data(r, 0) + q.states));
              }
            };
            fill(q.Dvectors, D);
            fill(q.Bvectors, B);
            fill(q.Uvectors, U);
            fill(q.rowRatioVectors, RR);
            q.compressedEmissionTable = asMat(compressedEmission);
            q.classicEmissionTable = asMat(classicEmission);
            q.foldedAscertainedCSFSmap = asCube(foldedAscertainedCSFS);
            q.ascertainedCSFSmap = asCube(ascertainedCSFS);
            using FArr = py::array_t<float, py::array::c_style | py::array::forcecast>;
            if (!CSFS.is_none()) {
              q.CSFSmap = asCube(CSFS.cast<FArr>());
            }
            if (!foldedCSFS.is_none()) {
              q.foldedCSFSmap = asCube(foldedCSFS.cast<FArr>());
            }
            if (!homozygousKeys.is_none() && !homozygousEmissions.is_none()) {
              const auto hk = homozygousKeys.cast<py::array_t<int32_t, py::array::c_style | py::array::forcecast>>();
              const auto hv = homozygousEmissions.cast<FArr>();
              if (hv.ndim() != 2 || hv.shape(0) != hk.size() || hv.shape(1) != static_cast<py::ssize_t>(q.states)) {
                throw std::runtime_error("homozygous emissions must be [keys][states]");
              }
              for (py::ssize_t r = 0; r < hk.size(); ++r) {
                q.homozygousEmissionMap[hk.at(r)] = std::vector<float>(hv.data(r, 0), hv.data(r, 0) + q.states);
              }
            }
            return q;
          },
          "CSFSSamples"_a, "discretization"_a, "expectedTimes"_a, "initialStateProb"_a, "columnRatios"_a, "keys"_a,
          "D"_a, "B"_a, "U"_a, "RR"_a, "compressedEmissionTable"_a, "classicEmissionTable"_a,
          "foldedAscertainedCSFSmap"_a, "ascertainedCSFSmap"_a, "CSFSmap"_a = py::none(),
          "foldedCSFSmap"_a = py::none(), "homozygousKeys"_a = py::none(), "homozygousEmissions"_a = py::none(),
          "Build decoding quantities from arrays instead of a file (synthetic models).")
      .def_readwrite("CSFSSamples", &DecodingQuantities::CSFSSamples)
      .def_readwrite("states", &DecodingQuantities::states)
      .def_readwrite("initialStateProb", &DecodingQuantities::initialStateProb)
      .def_readwrite("expectedTimes", &DecodingQuantities::expectedTimes)
      .def_readwrite("discretization", &DecodingQuantities::discretization)
      .def_readwrite("timeVector", &DecodingQuantities::timeVector)
      .def_readwrite("columnRatios", &DecodingQuantities::columnRatios)
      .def_readwrite("classicEmissionTable", &DecodingQuantities::classicEmissionTable)
      .def_readwrite("compressedEmissionTable", &DecodingQuantities::compressedEmissionTable)
      .def_property_readonly("Dvectors", [](const DecodingQuantities& q) { return keyedTableToDict(q.Dvectors, q.states); })
      .def_property_readonly("Bvectors", [](const DecodingQuantities& q) { return keyedTableToDict(q.Bvectors, q.states); })
      .def_property_readonly("Uvectors", [](const DecodingQuantities& q) { return keyedTableToDict(q.Uvectors, q.states); })
      .def_property_readonly("rowRatioVectors", [](const DecodingQuantities& q) { return keyedTableToDict(q.rowRatioVectors, q.states); })
      .def_readwrite("homozygousEmissionMap", &DecodingQuantities::homozygousEmissionMap)
      .def_readwrite("CSFSmap", &DecodingQuantities::CSFSmap)
      .def_readwrite("foldedCSFSmap", &DecodingQuantities::foldedCSFSmap)
      .def_readwrite("ascertainedCSFSmap", &DecodingQuantities::ascertainedCSFSmap)
      .def_readwrite("foldedAscertainedCSFSmap", &DecodingQuantities::foldedAscertainedCSFSmap);

  py::class_<DecodingParams>(m, "DecodingParams")
      .def(py::init<std::string, std::string, std::string, int, int, std::string, bool, bool, bool, bool, float, bool,
                    bool, bool, std::string, bool, bool>(),
           "inFileRoot"_a, "decodingQuantFile"_a, "outFileRoot"_a = "", "jobs"_a = 1, "jobInd"_a = 1,
           "decodingModeString"_a = "array", "decodingSequence"_a = false, "usingCSFS"_a = true, "compress"_a = false,
           "useAncestral"_a = false, "skipCSFSdistance"_a = 0.f, "noBatches"_a = false, "doPosteriorSums"_a = false,
           "doPerPairPosteriorMean"_a = false, "expectedCoalTimesFile"_a = "", "withinOnly"_a = false,
           "doMajorMinorPosteriorSums"_a = false)
      .def(py::init<>())
      .def(py::init<std::string, std::string, std::string, bool>(), "in_dir"_a, "decoding_quants"_a, "out_dir"_a,
           "FastSMC"_a = true)
      .def("validateParamsFastSMC", &DecodingParams::validateParamsFastSMC)
      .def("processOptions", &DecodingParams::processOptions)
      .def_readwrite("inFileRoot", &DecodingParams::inFileRoot)
      .def_readwrite("decodingQuantFile", &DecodingParams::decodingQuantFile)
      .def_readwrite("outFileRoot", &DecodingParams::outFileRoot)
      .def_readwrite("jobs", &DecodingParams::jobs)
      .def_readwrite("jobInd", &DecodingParams::jobInd)
      .def_readwrite("decodingModeString", &DecodingParams::decodingModeString)
      .def_readwrite("decodingMode", &DecodingParams::decodingMode)
      .def_readwrite("decodingModeOverall", &DecodingParams::decodingModeOverall)
      .def_readwrite("decodingSequence", &DecodingParams::decodingSequence)
      .def_readwrite("foldData", &DecodingParams::foldData)
      .def_readwrite("usingCSFS", &DecodingParams::usingCSFS)
      .def_readwrite("compress", &DecodingParams::compress)
      .def_readwrite("useAncestral", &DecodingParams::useAncestral)
      .def_readwrite("skipCSFSdistance", &DecodingParams::skipCSFSdistance)
      .def_readwrite("noBatches", &DecodingParams::noBatches)
      .def_readwrite("batchSize", &DecodingParams::batchSize)
      .def_readwrite("recallThreshold", &DecodingParams::recallThreshold)
      .def_readwrite("skip", &DecodingParams::skip)
      .def_readwrite("gap", &DecodingParams::gap)
      .def_readwrite("max_seeds", &DecodingParams::max_seeds)
      .def_readwrite("min_maf", &DecodingParams::min_maf)
      .def_readwrite("min_m", &DecodingParams::min_m)
      .def_readwrite("hashing", &DecodingParams::hashing)
      .def_readwrite("FastSMC", &DecodingParams::FastSMC)
      .def_readwrite("BIN_OUT", &DecodingParams::BIN_OUT)
      .def_readwrite("useKnownSeed", &DecodingParams::useKnownSeed)
      .def_readwrite("outputIbdSegmentLength", &DecodingParams::outputIbdSegmentLength)
      .def_readwrite("hashingWordSize", &DecodingParams::hashingWordSize)
      .def_readwrite("constReadAhead", &DecodingParams::constReadAhead)
      .def_readwrite("haploid", &DecodingParams::haploid)
      .def_readwrite("time", &DecodingParams::time)
      .def_readwrite("noConditionalAgeEstimates", &DecodingParams::noConditionalAgeEstimates)
      .def_readwrite("doPosteriorSums", &DecodingParams::doPosteriorSums)
      .def_readwrite("doPerPairMAP", &DecodingParams::doPerPairMAP)
      .def_readwrite("doPerPairPosteriorMean", &DecodingParams::doPerPairPosteriorMean)
      .def_readwrite("expectedCoalTimesFile", &DecodingParams::expectedCoalTimesFile)
      .def_readwrite("withinOnly", &DecodingParams::withinOnly)
      .def_readwrite("doMajorMinorPosteriorSums", &DecodingParams::doMajorMinorPosteriorSums)
      .def_readwrite("gpuDevice", &DecodingParams::gpuDevice);

  py::class_<IbdPairDataLine>(m, "IbdPairDataLine")
      .def(py::init<>())
      .def_readwrite("ind1FamId", &IbdPairDataLine::ind1FamId)
      .def_readwrite("ind1Id", &IbdPairDataLine::ind1Id)
      .def_readwrite("ind1Hap", &IbdPairDataLine::ind1Hap)
      .def_readwrite("ind2FamId", &IbdPairDataLine::ind2FamId)
      .def_readwrite("ind2Id", &IbdPairDataLine::ind2Id)
      .def_readwrite("ind2Hap", &IbdPairDataLine::ind2Hap)
      .def_readwrite("chromosome", &IbdPairDataLine::chromosome)
      .def_readwrite("ibdStart", &IbdPairDataLine::ibdStart)
      .def_readwrite("ibdEnd", &IbdPairDataLine::ibdEnd)
      .def_readwrite("lengthInCentimorgans", &IbdPairDataLine::lengthInCentimorgans)
      .def_readwrite("ibdScore", &IbdPairDataLine::ibdScore)
      .def_readwrite("postEst", &IbdPairDataLine::postEst)
      .def_readwrite("mapEst", &IbdPairDataLine::mapEst)
      .def("toString", &IbdPairDataLine::toString);

  py::class_<BinaryDataReader>(m, "BinaryDataReader")
      .def(py::init<const std::string&>(), "binaryFile"_a)
      .def("getNextLine", &BinaryDataReader::getNextLine)
      .def("moreLinesInFile", &BinaryDataReader::moreLinesInFile);

  py::class_<Data>(m, "Data")
      .def(py::init<const DecodingParams&>(), "params"_a)
      .def_static(
          "from_arrays",
          [](py::array_t<uint8_t, py::array::c_style | py::array::forcecast> alleles,
             py::array_t<int64_t, py::array::c_style | py::array::forcecast> bp,
             py::array_t<double, py::array::c_style | py::array::forcecast> cm, bool fold, bool knownSeed, int chr) {
            if (alleles.ndim() != 2 || bp.size() != alleles.shape(1) || cm.size() != alleles.shape(1)) {
              throw std::runtime_error("alleles must be [haplotypes][sites]; bp and cm one entry per site");
            }
            return Data::fromArrays(alleles.data(), static_cast<size_t>(alleles.shape(0)),
                                    static_cast<size_t>(alleles.shape(1)), bp.data(), cm.data(), fold, knownSeed, chr);
          },
          "alleles"_a, "bp"_a, "cm"_a, "foldToMinorAlleles"_a = true, "useKnownSeed"_a = true, "chrNumber"_a = 1)
      .def_static("countHapLines", &Data::countHapLines)
      .def_readwrite("FamIDList", &Data::FamIDList)
      .def_readwrite("IIDList", &Data::IIDList)
      .def_readwrite("famAndIndNameList", &Data::famAndIndNameList)
      .def_readwrite("sampleSize", &Data::sampleSize)
      .def_readwrite("haploidSampleSize", &Data::haploidSampleSize)
      .def_readwrite("sites", &Data::sites)
      .def_readwrite("decodingUsesCSFS", &Data::decodingUsesCSFS)
      .def_readwrite("geneticPositions", &Data::geneticPositions)
      .def_readwrite("physicalPositions", &Data::physicalPositions)
      .def_readwrite("siteWasFlippedDuringFolding", &Data::siteWasFlippedDuringFolding)
      .def_readwrite("recRateAtMarker", &Data::recRateAtMarker)
      .def_readwrite("derivedAlleleCounts", &Data::derivedAlleleCounts)
      .def_readwrite("totalSamplesCount", &Data::totalSamplesCount)
      .def_readwrite("chrNumber", &Data::chrNumber)
      .def_readwrite("windowSize", &Data::windowSize)
      .def_readwrite("w_i", &Data::w_i)
      .def_readwrite("w_j", &Data::w_j)
      .def_property_readonly("individuals", &Data::individuals,
                             "list of Individual (genotype1 / genotype2), unpacked from the bit matrix (Data.hpp:36)")
      .def_readonly("globalIndIndex", &Data::globalIndIndex, "sample-file line of every loaded individual")
      .def("calculateUndistinguishedCounts", &Data::calculateUndistinguishedCounts, "numCsfsSamples"_a,
           "Data::calculateUndistinguishedCounts (Data.cpp:567-599): [sites][3]")
      .def("genotype", &Data::genotypeVector, "hapRow"_a, "folded genotype of haplotype row 2*ind + (hap-1)")
      .def("packed_bits", [](const Data& d) {
        return toArray<uint64_t>(d.bits, {static_cast<py::ssize_t>(d.numHapRows()), static_cast<py::ssize_t>(d.wordsPerHap)});
      });

  py::class_<Individual>(m, "Individual")
      .def(py::init<int>(), "numOfSites"_a = 0)
      .def("setGenotype", &Individual::setGenotype, "hap"_a, "pos"_a, "val"_a)
      .def_readwrite("genotype1", &Individual::genotype1)
      .def_readwrite("genotype2", &Individual::genotype2);

  py::class_<HMM>(m, "HMM")
      .def(py::init([](const Data& d, const DecodingParams& p, int scalingSkip) { return new HMM(d, p, scalingSkip); }),
           "data"_a, "params"_a, "scalingSkip"_a = 1)
      .def(py::init([](const Data& d, const DecodingQuantities& q, const DecodingParams& p, int scalingSkip) {
             return new HMM(d, q, p, scalingSkip);
           }),
           "data"_a, "decodingQuantities"_a, "params"_a, "scalingSkip"_a = 1)
      .def("decode", py::overload_cast<const PairObservations&>(&HMM::decode))
      .def("decode", py::overload_cast<const PairObservations&, unsigned, unsigned>(&HMM::decode))
      .def("decodeAll", &HMM::decodeAll, "jobs"_a, "jobInd"_a)
      .def("pairsOfJob", &HMM::pairsOfJob, "jobs"_a, "jobInd"_a,
           "haplotype-row pairs decodeAll(jobs, jobInd) decodes, in its order (no decoding)")
      .def("getDecodingReturnValues", &HMM::getDecodingReturnValues, py::return_value_policy::reference_internal)
      .def("getDecodePairsReturnStruct", &HMM::getDecodePairsReturnStruct, py::return_value_policy::reference_internal)
      .def("decodePair", &HMM::decodePair)
      .def("decodePairs", &HMM::decodePairs)
      .def("decodeHapPair", &HMM::decodeHapPair)
      .def("decodeHapPairs", &HMM::decodeHapPairs)
      .def("decodeFromHashing", &HMM::decodeFromHashing, "hapA"_a, "hapB"_a, "fromPosition"_a, "toPosition"_a)
      .def("setShard", &HMM::setShard, "rank"_a, "world"_a,
           "decode only shard `rank` of `world` (contiguous whole batches) and write <file>.part<rank>of<world>")
      .def("ibdFileName", &HMM::ibdFileName, "jobs"_a, "jobInd"_a)
      .def("decodeSummarize", &HMM::decodeSummarize, "(MAP, posterior mean) per site of one pair (HMM.cpp:1498)")
      .def("getBatchBuffer", &HMM::getBatchBuffer,
           "PairObservations of the open batch; empty whenever a batch has just filled up (HMM.hpp:215)")
      .def("getQueuedPairs", &HMM::getQueuedPairs, "pairs waiting for the next launch, full batches included")
      .def("finishDecoding", &HMM::finishDecoding)
      .def("setWritePerPairPosteriorMean", &HMM::setWritePerPairPosteriorMean, "writePerPairPosteriorMean"_a = true,
           "ASMC mode: write <outFileRoot>.perPairPosteriorMeans.gz, one row per decoded pair (HMM.hpp:287, HMM.cpp:1412-1416)")
      .def("setWritePerPairMap", &HMM::setWritePerPairMap, "writePerPairMAP"_a = true,
           "ASMC mode: write <outFileRoot>.perPairMAP.gz, one row per decoded pair (HMM.hpp:293, HMM.cpp:1417-1420)")
      .def("setStorePerPairPosteriorMean", &HMM::setStorePerPairPosteriorMean, "storePerPairPosteriorMean"_a = true)
      .def("setStorePerPairMap", &HMM::setStorePerPairMap, "storePerPairMAP"_a = true)
      .def("getExpectedCoalTimes", &HMM::getExpectedCoalTimes,
           "expected coalescence times the per-pair posterior means use: the intervals file's second column when "
           "DecodingParams.expectedCoalTimesFile names one, else the decoding quantities' (HMM.cpp:1736-1748)")
      .def("finishFromHashing", &HMM::finishFromHashing)
      .def("closeIBDFile", &HMM::closeIBDFile)
      .def("getDecodingQuantities", &HMM::getDecodingQuantities, py::return_value_policy::reference_internal)
      .def("makePairObs", &HMM::makePairObs, "iHap"_a, "ind1"_a, "jHap"_a, "ind2"_a)
      .def("setKeepIbdRecords", &HMM::setKeepIbdRecords)
      .def("getNumSegmentsDetected", &HMM::getNumSegmentsDetected)
      .def("getIbdRecords",
           [](const HMM& h) {
             // (hapA, hapB, start, end, prob, postMean, map) per record, in output order
             py::list out;
             const auto& r = h.getIbdRecords();
             const auto& p = h.getIbdRecordPairs();
             for (size_t i = 0; i < r.size(); ++i) {
               out.append(py::make_tuple(p[i].hap_a, p[i].hap_b, r[i].start, r[i].end, r[i].prob, r[i].post_mean, r[i].map));
             }
             return out;
           })
      .def("setWriteIbdFile", &HMM::setWriteIbdFile, "write"_a,
           "False: open no IBD file, only keep the records (setKeepIbdRecords) -- multi-GPU ranks whose records are gathered")
      .def("writeIbdRecordArrays",
           [](const HMM& h, const std::string& fileName, py::array_t<uint32_t, py::array::c_style | py::array::forcecast> ha,
              py::array_t<uint32_t, py::array::c_style | py::array::forcecast> hb,
              py::array_t<int32_t, py::array::c_style | py::array::forcecast> st,
              py::array_t<int32_t, py::array::c_style | py::array::forcecast> en,
              py::array_t<float, py::array::c_style | py::array::forcecast> pr,
              py::array_t<float, py::array::c_style | py::array::forcecast> pm,
              py::array_t<float, py::array::c_style | py::array::forcecast> mp) {
             // the IBD file of records given as columns (those of getIbdRecordArrays, e.g. gathered from several ranks)
             const size_t n = static_cast<size_t>(ha.size());
             for (const py::ssize_t other : {hb.size(), st.size(), en.size(), pr.size(), pm.size(), mp.size()}) {
               if (static_cast<size_t>(other) != n) {
                 throw std::runtime_error("writeIbdRecordArrays: columns of different lengths");
               }
             }
             std::vector<fsmc_pair> pairs(n);
             std::vector<fsmc_ibd_record> recs(n);
             for (size_t i = 0; i < n; ++i) {
               const py::ssize_t k = static_cast<py::ssize_t>(i);
               pairs[i].hap_a = ha.at(k);
               pairs[i].hap_b = hb.at(k);
               recs[i].pair = static_cast<uint32_t>(i);
               recs[i].start = st.at(k);
               recs[i].end = en.at(k);
               recs[i].prob = pr.at(k);
               recs[i].post_mean = pm.at(k);
               recs[i].map = mp.at(k);
             }
             h.writeIbdRecordsTo(fileName, pairs, recs);
           },
           "fileName"_a, "hap_a"_a, "hap_b"_a, "start"_a, "end"_a, "prob"_a, "post_mean"_a, "map"_a)
      .def("getIbdRecordArrays",
           [](const HMM& h) {
             // the kept records as columns (numpy): pair ordinal, hapA, hapB, start, end, prob, postMean, map
             const auto& r = h.getIbdRecords();
             const auto& p = h.getIbdRecordPairs();
             const auto& o = h.getIbdRecordOrdinals();
             const py::ssize_t n = static_cast<py::ssize_t>(r.size());
             py::array_t<uint64_t> ord(n);
             py::array_t<uint32_t> ha(n), hb(n);
             py::array_t<int32_t> st(n), en(n);
             py::array_t<float> pr(n), pm(n), mp(n);
             for (py::ssize_t i = 0; i < n; ++i) {
               const size_t k = static_cast<size_t>(i);
               ord.mutable_at(i) = k < o.size() ? o[k] : 0;
               ha.mutable_at(i) = p[k].hap_a;
               hb.mutable_at(i) = p[k].hap_b;
               st.mutable_at(i) = r[k].start;
               en.mutable_at(i) = r[k].end;
               pr.mutable_at(i) = r[k].prob;
               pm.mutable_at(i) = r[k].post_mean;
               mp.mutable_at(i) = r[k].map;
             }
             py::dict d;
             d["pair"] = ord;
             d["hap_a"] = ha;
             d["hap_b"] = hb;
             d["start"] = st;
             d["end"] = en;
             d["prob"] = pr;
             d["post_mean"] = pm;
             d["map"] = mp;
             return d;
           },
           "the kept IBD records as numpy columns; `pair` = ordinal of the record's pair among all pairs decoded so far")
      .def("getIbdLines",
           [](const HMM& h) {
             const auto& r = h.getIbdRecords();
             const auto& p = h.getIbdRecordPairs();
             return h.formatIbdRecords(p.data(), r.data(), r.size());
           })
      .def("preparedModel", [](const HMM& h) {
        const PreparedModel& pm = h.getPreparedModel();
        py::dict d;
        const py::ssize_t K = pm.K, S = pm.S, R = pm.nRows;
        d["K"] = pm.K;
        d["S"] = pm.S;
        d["pi"] = toArray<float>(pm.pi, {K});
        d["col_ratios"] = toArray<float>(pm.colRatios, {K});
        d["exp_times"] = toArray<float>(pm.expTimes, {K});
        d["D"] = toArray<float>(pm.D, {R, K});
        d["B"] = toArray<float>(pm.B, {R, K});
        d["U"] = toArray<float>(pm.U, {R, K});
        d["RR"] = toArray<float>(pm.RR, {R, K});
        d["step_row"] = toArray<int32_t>(pm.stepRow, {S});
        d["e1"] = toArray<float>(pm.e1, {S, K});
        d["e0m1"] = toArray<float>(pm.e0m1, {S, K});
        d["e2m0"] = toArray<float>(pm.e2m0, {S, K});
        d["state_threshold"] = pm.stateThreshold;
        d["age_threshold"] = pm.ageThreshold;
        d["probability_threshold"] = pm.probabilityThreshold;
        d["sequence"] = pm.sequence;
        if (pm.sequence) {
          d["gap_row_f"] = toArray<int32_t>(pm.gapRowF, {S});
          d["site_row_f"] = toArray<int32_t>(pm.siteRowF, {S});
          d["gap_row_b"] = toArray<int32_t>(pm.gapRowB, {S});
          d["site_row_b"] = toArray<int32_t>(pm.siteRowB, {S});
          d["hom"] = toArray<float>(pm.hom, {S, K});
        }
        return d;
      });

  py::class_<FastSMC>(m, "FastSMC")
      .def(py::init<DecodingParams>(), "decodingParams"_a)
      .def(py::init<const std::string&, const std::string&>(), "in_dir"_a, "out_dir"_a)
      .def("run", &FastSMC::run)
      .def("setShard", [](FastSMC& f, int rank, int world) { f.hmm().setShard(rank, world); }, "rank"_a, "world"_a)
      .def("outputFileName", &FastSMC::outputFileName)
      .def("hmm", &FastSMC::hmm, py::return_value_policy::reference_internal);

  py::class_<ASMC>(m, "ASMC")
      .def(py::init<DecodingParams>(), "decodingParams"_a)
      .def(py::init<const std::string&, const std::string&, const std::string&>(), "in_dir"_a, "dq_file"_a,
           "out_dir"_a = "")
      .def("decodeAllInJob", &ASMC::decodeAllInJob)
      .def("decodePairs",
           py::overload_cast<const std::vector<unsigned long>&, const std::vector<unsigned long>&, bool, bool, bool,
                             bool>(&ASMC::decodePairs),
           "hap_indices_a"_a, "hap_indices_b"_a, "per_pair_posteriors"_a = false, "sum_of_posteriors"_a = false,
           "per_pair_posterior_means"_a = false, "per_pair_MAPs"_a = false)
      .def("decodePairs",
           py::overload_cast<const std::vector<std::string>&, const std::vector<std::string>&, bool, bool, bool, bool>(
               &ASMC::decodePairs),
           "hap_ids_a"_a, "hap_ids_b"_a, "per_pair_posteriors"_a = false, "sum_of_posteriors"_a = false,
           "per_pair_posterior_means"_a = false, "per_pair_MAPs"_a = false)
      .def("get_copy_of_results", &ASMC::getCopyOfResults, py::return_value_policy::copy)
      .def("get_ref_of_results", &ASMC::getRefOfResults, py::return_value_policy::reference_internal)
      .def("hmm", &ASMC::hmm, py::return_value_policy::reference_internal);

  m.def("cmBetween", &cmBetween, "w1"_a, "w2"_a, "geneticPositions"_a, "wordSize"_a);
  py::class_<Match>(m, "Match")
      .def(py::init<unsigned long, int>(), "wordSize"_a = 64, "i"_a = 0)
      .def("extend", &Match::extend)
      .def("addGap", &Match::addGap)
      .def("getGaps", &Match::getGaps)
      .def("getWordSize", &Match::getWordSize)
      .def("getInterval", [](const Match& x) { return std::vector<int>{x.start(), x.end()}; });
  m.def("hashingCandidates",
        [](const Data& d, const DecodingParams& p) {
          py::list out;
          for (const auto& c : hashingCandidates(d, p)) {
            out.append(py::make_tuple(c.hapA, c.hapB, c.from, c.to));
          }
          return out;
        },
        "data"_a, "params"_a,
        "candidate (hapA, hapB, fromSite, toSite) list of the identification step from the HOST restatement of the "
        "reference's hash maps -- the checker of hashingCandidatesDevice in the tests; FastSMC.run() does not use it");
  m.def("hashingWords",
        [](const Data& d, const DecodingParams& p) {
          const HashingPrefilter pf(d, p);
          return toArray<uint64_t>(pf.words(), {static_cast<py::ssize_t>(pf.numHaps()),
                                                static_cast<py::ssize_t>(pf.numWords())});
        },
        "data"_a, "params"_a,
        "the hashing words of every haplotype row, [haps][words] uint64: what Individuals::getWordHash returns for "
        "word w of the haplotype (Individuals.hpp:46-59) -- hashingWordSize sites to a word, MAF-filtered sites skipped");
  m.def("hashingCandidatesDevice",
        [](const Data& d, const DecodingParams& p, int device) {
          py::list out;
          for (const auto& c : hashingCandidatesDevice(d, p, device)) {
            out.append(py::make_tuple(c.hapA, c.hapB, c.from, c.to));
          }
          return out;
        },
        "data"_a, "params"_a, "device"_a = 0,
        "candidate (hapA, hapB, fromSite, toSite) list of the identification step, computed on the GPU (fsmc_identify) "
        "-- what FastSMC.run() hands to HMM.decodeFromHashing");
  // StringUtils::stof / stod (StringUtils.cpp:36-44): std::stold narrowed; std::out_of_range -> OverflowError is not a
  // pybind default, so both standard exceptions surface as ValueError subclasses with the C++ type in the message
  m.def("stof", [](const std::string& s) {
    try {
      return refStof(s);
    } catch (const std::out_of_range& e) {
      throw py::value_error(std::string("std::out_of_range: ") + e.what());
    } catch (const std::invalid_argument& e) {
      throw py::value_error(std::string("std::invalid_argument: ") + e.what());
    }
  });
  m.def("stod", [](const std::string& s) {
    try {
      return refStod(s);
    } catch (const std::out_of_range& e) {
      throw py::value_error(std::string("std::out_of_range: ") + e.what());
    } catch (const std::invalid_argument& e) {
      throw py::value_error(std::string("std::invalid_argument: ") + e.what());
    }
  });
  m.def("roundMorgans", &roundMorgans, "value"_a, "precision"_a, "min"_a);
  m.def("roundPhysical", &roundPhysical, "value"_a, "precision"_a);
  m.def("getFromPosition", &getFromPosition, "geneticPositions"_a, "from"_a, "cmDist"_a = 0.5f);
  m.def("getToPosition", &getToPosition, "geneticPositions"_a, "to"_a, "cmDist"_a = 0.5f);
  m.def("hapToDipId", &hapToDipId);
  m.def("dipToHapId", &dipToHapId);
  m.def("indPlusHapToCombinedId", &indPlusHapToCombinedId);
  m.def("combinedIdToIndPlusHap", &combinedIdToIndPlusHap);
  fsmc_host::bindContainers(m);
}
