import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    print(f, 'headline ms %.4f'%d['ms_per_step'], 'c3 b4096 %.4f sat %.4f'%(d['config3']['b4096']['ms_per_step'], d['config3']['saturating']['ms_per_step']), 'c4 %.4f'%d['config4']['seconds'], 'c5 %.4f'%d['config5']['ms_per_step'], 'qp %.2f %.2f'%(d['qp_solve']['snap8']['ms_per_batch'], d['qp_solve']['jerk5']['ms_per_batch']), 'b1024 %.4f graph %.4f'%(d['config1_b1024']['ms_per_step'], d['config1_b1024']['graph64x8']['ms_per_launch']), 'sampler %.4f'%d['config1_b1024']['sampler']['one_problem']['ms_per_launch'])
