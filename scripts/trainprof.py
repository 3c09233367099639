import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = ['trainbench.py', sys.argv[1], sys.argv[2]]
pr = cProfile.Profile()
src = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trainbench.py')).read()
code = compile(src, 'trainbench.py', 'exec')
g = {'__name__': '__main__', '__file__': os.path.join(os.path.dirname(os.path.abspath(__file__)), 'trainbench.py')}
pr.enable(); exec(code, g); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('cumulative').print_stats(45); print(s.getvalue()[:9000])
