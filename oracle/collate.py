"""TEST INFRASTRUCTURE — numpy restatement of the reference's batch assembly.

`collate` follows LargeConformerBasedDataset.collate_fn (conan_fgw/src/data/datasets.py:170-199): the K conformer Data objects
of every molecule are concatenated molecule-major by PyG's Batch.from_data_list (node tensors concatenated, edge_index shifted
by the number of nodes before each graph, `batch` = graph id per node [3P-memory: PyG 2.3.0, not under /root/reference]) and
`batch_node_index` repeats the conformer-graph id once per atom (:183-184, :198).  `aggregation_index` follows
EquivAggregation.create_aggregation_index (conan_fgw/src/model/common.py:414-423).  PARITY UNPINNED: the reference holds no
fixture at this boundary (its dataset files are not shipped)."""
import numpy as np


def collate(items, K):
    z, pos, x, batch, ei, ea, y, smiles, bni, cnb = [], [], [], [], [], [], [], [], [], []
    node0, mol_id, node_count = 0, -1, 0
    for it in items:                                   # datasets.py:180-184: one id per conformer graph, molecule-major
        n = len(it.z)
        for k in range(K):
            mol_id += 1
            z.append(np.asarray(it.z, np.int64)); pos.append(np.asarray(it.pos[k], np.float32)); x.append(np.asarray(it.x, np.float32))
            batch.append(np.full(n, mol_id, np.int64)); bni.append(np.full(n, mol_id, np.int64))
            ei.append(np.asarray(it.edge_index, np.int64) + node0); ea.append(np.asarray(it.edge_attr, np.float32))
            y.append(np.float32(it.y)); smiles.append(it.smiles)
            node0 += n
        cnb.append(np.tile(np.arange(n, dtype=np.int64) + node_count, K))      # datasets.py:189-192: (arange(n) + node_count).repeat(num_confs)
        node_count += n
    cat = lambda a, ax=0: np.concatenate(a, axis=ax)
    return dict(z=cat(z), pos=cat(pos), x=cat(x), batch=cat(batch), batch_node_index=cat(bni), edge_index=cat(ei, 1), edge_attr=cat(ea),
                y=np.asarray(y, np.float32), smiles=smiles, conf_node_batch=cat(cnb))


def aggregation_index(smiles, K):
    index, mol_idx, i = [], -1, 0                      # common.py:414-423, loop kept as written there
    while i < len(smiles):
        mol_idx += 1
        for _ in range(K):
            index.append(mol_idx)
            i += 1
    return np.asarray(index, np.int64)
