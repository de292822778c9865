"""Pins oracle/ec.py (the restated `fastecdsa` boundary) to PUBLIC secp256k1
known answers (SEC 2 v2 sec. 2.4.1 constants; the widely published k*G table)."""
import pytest

from oracle.ec import INF, Point, mod_sqrt, secp256k1, point_from_le64, point_to_le64

G = secp256k1.G
KG = {
    1: (0x79BE667EF9DCBBAC55A06295CE870B07029BFCDB2DCE28D959F2815B16F81798,
        0x483ADA7726A3C4655DA4FBFC0E1108A8FD17B448A68554199C47D08FFB10D4B8),
    2: (0xC6047F9441ED7D6D3045406E95C07CD85C778E4B8CEF3CA7ABAC09B95C709EE5,
        0x1AE168FEA63DC339A3C58419466CEAEEF7F632653266D0E1236431A950CFE52A),
    3: (0xF9308A019258C31049344F85F89D5229B531C845836F99B08601F113BCE036F9,
        0x388F7B0F632DE8140FE337E62A37F3566500A99934C2231B6CB9FD7584B8E672),
}


def test_constants():
    assert secp256k1.p == 2**256 - 2**32 - 977
    assert secp256k1.q == 0xFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFFEBAAEDCE6AF48A03BBFD25E8CD0364141
    assert secp256k1.is_point_on_curve((G.x, G.y))


@pytest.mark.parametrize("k", sorted(KG))
def test_known_multiples(k):
    assert ((k * G).x, (k * G).y) == KG[k]
    acc = INF
    for _ in range(k):
        acc = acc + G
    assert (acc.x, acc.y) == KG[k]


def test_group_law_edges():
    q = secp256k1.q
    assert (q - 1) * G == -G
    assert q * G == INF and 0 * G == INF
    assert G + (-G) == INF
    assert G + INF == G and INF + G == G and INF + INF == INF
    assert G + G == 2 * G
    assert (q + 5) * G == 5 * G and (-3) * G == -(3 * G)
    a, b, c = 12345 * G, 67890 * G, 424242 * G
    assert (a + b) + c == a + (b + c) and a + b == b + a
    # 7 is a non-residue, so (0,0) is a safe identity encoding
    assert pow(7, (secp256k1.p - 1) // 2, secp256k1.p) == secp256k1.p - 1
    with pytest.raises(ValueError):
        Point(1, 1, secp256k1)


def test_mod_sqrt_and_layout():
    p = secp256k1.p
    r, r2 = mod_sqrt(G.y * G.y % p, p)
    assert {r, r2} == {G.y, p - G.y} and r == pow(G.y * G.y % p, (p + 1) // 4, p)
    assert point_from_le64(point_to_le64(3 * G)) == 3 * G
    assert point_to_le64(INF) == bytes(64) and point_from_le64(bytes(64)) == INF
