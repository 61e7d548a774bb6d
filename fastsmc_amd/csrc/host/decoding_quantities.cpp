// decoding_quantities.cpp -- parser for *.decodingQuantities.gz (format: SURVEY.md App. B;
// reference parser: DecodingQuantities.cpp:60-345; writer: TOOLS/.../DecodingQuantities.java:190-299).
#include "decoding_quantities.hpp"

#include <stdexcept>

#include "util.hpp"

namespace fsmc_host
{

void KeyedTable::add(float key, const std::vector<float>& row)
{
  if (key == 0.f) {
    key = 0.f; // -0 and +0 are one key for unordered_map<float>
  }
  auto it = index.find(floatBits(key));
  if (it != index.end()) { // a repeated key overwrites, like operator[] in the reference
    std::copy(row.begin(), row.end(), values.begin() + static_cast<size_t>(it->second) * row.size());
    return;
  }
  index.emplace(floatBits(key), static_cast<int>(keys.size()));
  keys.push_back(key);
  values.insert(values.end(), row.begin(), row.end());
}

int KeyedTable::find(float key) const
{
  if (key == 0.f) {
    key = 0.f;
  }
  auto it = index.find(floatBits(key));
  return it == index.end() ? -1 : it->second;
}

DecodingQuantities::DecodingQuantities(const std::string& fileName)
{
  validateDecodingQuantitiesFile(fileName);
  parse(fileName);
}

void DecodingQuantities::validateDecodingQuantitiesFile(const std::string& fileName)
{
  if (!fileExists(fileName)) {
    throw std::runtime_error("ERROR: Decoding quantities file " + fileName + " does not exist.\n");
  }
  LineReader br(fileName);
  std::string firstLine;
  br.getline(firstLine);
  if (firstLine != "TransitionType") {
    throw std::runtime_error("ERROR: Decoding quantities file " + fileName +
                             " does not seem to contain the correct information.\n"
                             "Expected file to begin with \"TransitionType\", but instead found \"" +
                             firstLine + "\"\n");
  }
}

namespace
{
enum class Section { None, ColumnRatios, InitialStateProb, RowRatios, Uvectors, Bvectors, Dvectors, Homozygous };

std::vector<float> parseFloats(const std::vector<std::string>& tok, size_t first = 0)
{
  std::vector<float> out;
  out.reserve(tok.size() - first);
  for (size_t i = first; i < tok.size(); ++i) {
    out.push_back(refStof(tok[i]));
  }
  return out;
}
} // namespace

void DecodingQuantities::parse(const std::string& fileName)
{
  LineReader br(fileName);
  std::string line;
  Section section = Section::None;
  bool haveStates = false, haveSamples = false;

  auto nextLine = [&]() -> std::string {
    std::string l;
    br.getline(l);
    return l;
  };
  auto requireStates = [&](const char* what) {
    if (!haveStates) {
      throw std::runtime_error(std::string("ERROR. Parsed ") + what + " before parsing states.");
    }
  };
  auto readRow = [&](const char* what) -> std::vector<float> {
    std::vector<float> v = parseFloats(splitWhitespace(nextLine()));
    if (v.size() != states) {
      throw std::runtime_error(std::string("ERROR. Parsed ") + std::to_string(v.size()) + " " + what +
                               " entries for " + std::to_string(states) + " states.");
    }
    return v;
  };
  auto readBlock = [&](const char* what, int rows) {
    std::vector<std::vector<float>> out;
    for (int r = 0; r < rows; ++r) {
      out.push_back(readRow(what));
    }
    return out;
  };
  auto csfsSlot = [&](std::vector<std::vector<std::vector<float>>>& map, const std::vector<std::string>& tok,
                      const char* what) -> std::vector<std::vector<float>>& {
    if (!haveStates || !haveSamples) {
      throw std::runtime_error(std::string("ERROR. Parsed ") + what +
                               " before parsing states and number of CSFS samples.");
    }
    if (tok.size() < 2) {
      throw std::runtime_error(std::string("ERROR. ") + what + " header without an index.");
    }
    const int u = std::stoi(tok[1]);
    if (u < 0 || static_cast<size_t>(u) >= map.size()) {
      throw std::runtime_error(std::string("ERROR. ") + what + " index " + tok[1] + " out of range.");
    }
    return map[static_cast<size_t>(u)];
  };

  while (br.getline(line)) {
    const std::vector<std::string> tok = splitWhitespace(line);
    if (tok.empty()) {
      continue;
    }
    const std::string head = toLower(tok[0]);
    if (head == "states") {
      states = static_cast<unsigned>(std::stoi(nextLine()));
      haveStates = true;
    } else if (head == "transitiontype") {
      nextLine();
    } else if (head == "csfssamples") {
      CSFSSamples = std::stoi(nextLine());
      haveSamples = true;
      const size_t n = static_cast<size_t>(CSFSSamples - 1);
      CSFSmap.assign(n, {});
      foldedCSFSmap.assign(n, {});
      ascertainedCSFSmap.assign(n, {});
      foldedAscertainedCSFSmap.assign(n, {});
    } else if (head == "timevector") {
      timeVector = parseFloats(splitWhitespace(nextLine()));
    } else if (head == "sizevector") {
      nextLine();
    } else if (head == "expectedtimes") {
      requireStates("ExpectedTimes");
      expectedTimes = readRow("ExpectedTimes");
    } else if (head == "discretization") {
      requireStates("Discretization");
      discretization = parseFloats(splitWhitespace(nextLine()));
      if (discretization.size() != states + 1) {
        throw std::runtime_error("ERROR. Parsed " + std::to_string(discretization.size()) +
                                 " Discretization entries for " + std::to_string(states) + " states.");
      }
    } else if (head == "classicemission") {
      requireStates("ClassicEmission");
      classicEmissionTable = readBlock("ClassicEmission", 2);
    } else if (head == "compressedascertainedemission") {
      requireStates("CompressedAscertainedEmission");
      compressedEmissionTable = readBlock("CompressedAscertainedEmission", 2);
    } else if (head == "csfs") {
      csfsSlot(CSFSmap, tok, "CSFS") = readBlock("CSFS", 3);
    } else if (head == "foldedcsfs") {
      csfsSlot(foldedCSFSmap, tok, "FoldedCSFS") = readBlock("FoldedCSFS", 2);
    } else if (head == "ascertainedcsfs") {
      csfsSlot(ascertainedCSFSmap, tok, "AscertainedCSFS") = readBlock("AscertainedCSFS", 3);
    } else if (head == "foldedascertainedcsfs") {
      csfsSlot(foldedAscertainedCSFSmap, tok, "FoldedAscertainedCSFS") = readBlock("FoldedAscertainedCSFS", 2);
    } else if (head == "homozygousemissions") {
      section = Section::Homozygous;
    } else if (head == "initialstateprob") {
      section = Section::InitialStateProb;
    } else if (head == "columnratios") {
      section = Section::ColumnRatios;
    } else if (head == "rowratios") {
      section = Section::RowRatios;
    } else if (head == "uvectors") {
      section = Section::Uvectors;
    } else if (head == "bvectors") {
      section = Section::Bvectors;
    } else if (head == "dvectors") {
      section = Section::Dvectors;
    } else {
      // a content line of the current section; rows shorter than `states` are zero padded
      auto padded = [&](size_t first) {
        std::vector<float> row(states, 0.f);
        for (size_t i = first; i < tok.size() && i - first < states; ++i) {
          row[i - first] = refStof(tok[i]);
        }
        return row;
      };
      switch (section) {
      case Section::ColumnRatios:
        columnRatios = padded(0);
        break;
      case Section::InitialStateProb:
        initialStateProb = padded(0);
        break;
      case Section::RowRatios:
        rowRatioVectors.add(refStof(tok[0]), padded(1));
        break;
      case Section::Uvectors:
        Uvectors.add(refStof(tok[0]), padded(1));
        break;
      case Section::Bvectors:
        Bvectors.add(refStof(tok[0]), padded(1));
        break;
      case Section::Dvectors:
        Dvectors.add(refStof(tok[0]), padded(1));
        break;
      case Section::Homozygous:
        homozygousEmissionMap[std::stoi(tok[0])] = padded(1);
        break;
      case Section::None:
        break;
      }
    }
  }
  if (!haveStates || states < 2) {
    throw std::runtime_error("ERROR: decoding quantities file " + fileName + " has no States section.");
  }
}

} // namespace fsmc_host
