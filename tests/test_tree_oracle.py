"""Known answers of the token-tree oracle (no GPU)."""
import torch

from oracle import tree as otree
from oracle.attention import verify_attention


def test_build_tree_merges_common_prefixes_in_first_appearance_order():
    toks, par, dep = otree.build_tree([[10, 11, 12], [10, 13], [20], [10, 11, 14, 15]], k=3, max_nodes=15)
    assert toks == [0, 10, 11, 12, 13, 20, 14]
    assert par == [0, 0, 1, 2, 1, 0, 2]
    assert dep == [0, 1, 2, 3, 2, 1, 3]
    assert otree.ancestor_masks(par) == [0b1, 0b11, 0b111, 0b1111, 0b10011, 0b100001, 0b1000111]
    # the node budget cuts late candidates, never an earlier chain
    toks, par, _ = otree.build_tree([[1, 2, 3], [4, 5, 6]], k=3, max_nodes=4)
    assert toks == [0, 1, 2, 3, 4] and par == [0, 0, 1, 2, 0]


def test_accept_walk_known_answers():
    toks, par, _ = otree.build_tree([[10, 11, 12], [10, 13], [20]], k=3, max_nodes=15)
    # rows: 1=10, 2=11, 3=12, 4=13, 5=20
    assert otree.accept_greedy([10, 13, 0, 0, 99, 0], toks, par, 3) == ([1, 4], [10, 13, 99])
    assert otree.accept_greedy([20, 0, 0, 0, 0, 55], toks, par, 3) == ([5], [20, 55])
    assert otree.accept_greedy([7, 0, 0, 0, 0, 0], toks, par, 3) == ([], [7])
    assert otree.accept_greedy([10, 11, 12, 42, 0, 0], toks, par, 3) == ([1, 2, 3], [10, 11, 12, 42])
    assert otree.accept_greedy([10, 11, 12, 42, 0, 0], toks, par, 2) == ([1, 2], [10, 11, 12])


def test_tree_attention_of_a_chain_is_causal_attention():
    g = torch.Generator().manual_seed(0)
    n_ctx, T, Hq, Hkv, D = 9, 4, 4, 2, 16
    q = torch.randn(T, Hq, D, generator=g)
    k = torch.randn(n_ctx + T, Hkv, D, generator=g)
    v = torch.randn(n_ctx + T, Hkv, D, generator=g)
    masks = otree.ancestor_masks([0, 0, 1, 2])
    a = otree.tree_attention(q, k, v, n_ctx, masks, 0.25)
    b = verify_attention(q, k, v, [n_ctx + t for t in range(T)], 0.25)
    torch.testing.assert_close(a, b)
    # a sibling does not see its sibling: dropping the sibling's key changes nothing
    masks = otree.ancestor_masks([0, 0, 0, 1])          # rows 1 and 2 are siblings, 3 under 1
    full = otree.tree_attention(q, k, v, n_ctx, masks, 0.25)
    k2, v2 = k.clone(), v.clone()
    k2[n_ctx + 2], v2[n_ctx + 2] = 0.0, 0.0
    alt = otree.tree_attention(q, k2, v2, n_ctx, masks, 0.25)
    torch.testing.assert_close(full[[0, 1, 3]], alt[[0, 1, 3]])


def test_lookup_candidates_orders_by_ngram_length_then_recency():
    h = [1, 2, 3, 9, 1, 2, 4, 8, 2, 5, 7, 1, 2]
    # 2-gram (1, 2): matches end at 5 (-> 4, 8, 2) and 1 (-> 3, 9, 1); 1-gram 2: 8 (-> 5, 7, 1), ...
    assert otree.lookup_candidates(h, 3, 2, 1, 3) == [[4, 8, 2], [3, 9, 1], [5, 7, 1]]
    assert otree.lookup_candidates(h, 3, 2, 1, 1) == [[4, 8, 2]]
    assert otree.lookup_candidates([5], 3, 2, 1, 3) == []
