"""The container classes of the reference's Python module (pybind.cpp:63-70) exist under the same names, are constructible,
behave as sequences / mappings, and are accepted wherever the module takes a list."""
import numpy as np

from asmc import pyASMC as m


def test_vector_and_map_classes_of_the_reference_module():
    v = m.VectorFloat([1.0, 2.5])
    v.append(3.0)
    assert list(v) == [1.0, 2.5, 3.0] and len(v) == 3 and v[1] == 2.5
    b = m.VectorBool([True, False, True])
    assert list(b) == [True, False, True]
    u = m.VectorUInt()
    u.append(7)
    u.extend([8, 9])
    assert list(u) == [7, 8, 9]
    mat = m.Matrix()
    mat.append(m.VectorFloat([1, 2]))
    assert [list(r) for r in mat] == [[1.0, 2.0]]
    um = m.UMapFloatToVectorFloat()
    um[0.5] = m.VectorFloat([1, 2, 3])
    assert list(um.keys()) == [0.5] and list(um[0.5]) == [1.0, 2.0, 3.0] and 0.5 in um
    ui = m.UMapIntToVectorFloat()
    ui[3] = m.VectorFloat([4])
    assert {k: list(x) for k, x in ui.items()} == {3: [4.0]}
    vi = m.VectorIndividual()
    vi.append(m.Individual(4))
    assert len(vi) == 1 and len(vi[0].genotype1) == 4
    assert len(m.VectorPairObservations()) == 0


def test_module_functions_take_the_container_objects():
    gen = [0.0, 0.001, 0.002, 0.01, 0.02, 0.03]
    assert m.getFromPosition(m.VectorFloat(gen), 3) == m.getFromPosition(gen, 3)
    assert m.getToPosition(m.VectorFloat(gen), 3) == m.getToPosition(gen, 3)
    np.testing.assert_array_equal(np.array(m.VectorFloat(gen), np.float32), np.array(gen, np.float32))
